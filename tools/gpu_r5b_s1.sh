#!/bin/bash
# session 1 of the second half of round 5: the new parity tests (sampler variants, 7-column boxes, model_channels = 384) + the rows suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r5b_s1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_rows.py tests/test_hip_scene.py -x -q -m gpu -k "384 or sampler_variants or box_postprocess or layout_loop or unet1d or abi" > $OUT/new_tests.log 2>&1
echo "new tests rc=$?" > $OUT/summary.txt
tail -5 $OUT/new_tests.log >> $OUT/summary.txt
timeout 600 python bench.py --workload layout --steps 1000 --warmup 2 > $OUT/bench_layout.json 2> $OUT/bench_layout.err
tail -1 $OUT/bench_layout.json | cut -c1-400 >> $OUT/summary.txt
cat $OUT/summary.txt
