#!/bin/bash
# session 2: the folded self-attention (ES_PRO_LN_ATTN): rows suite, 1000-step trajectory, scene goldens, layout line (no CPU baseline)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r5b_s2
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_rows.py -x -q -m gpu > $OUT/rows_tests.log 2>&1
echo "rows tests rc=$?" > $OUT/summary.txt
tail -15 $OUT/rows_tests.log >> $OUT/summary.txt
timeout 600 python -m pytest tests/test_hip_traj.py -x -q -m gpu -k "layout" > $OUT/traj_tests.log 2>&1
echo "traj tests rc=$?" >> $OUT/summary.txt
tail -5 $OUT/traj_tests.log >> $OUT/summary.txt
timeout 400 python bench.py --workload layout --steps 1000 --warmup 2 --no-cpu-baseline --no-sub-records > $OUT/bench_layout.json 2> $OUT/bench_layout.err
tail -1 $OUT/bench_layout.json | cut -c1-500 >> $OUT/summary.txt
timeout 600 python -m pytest tests/test_hip_vol.py -x -q -m gpu -k "layernorm or groupnorm or unet3d_tiny or shards" > $OUT/vol_tests.log 2>&1
echo "vol tests rc=$?" >> $OUT/summary.txt
tail -5 $OUT/vol_tests.log >> $OUT/summary.txt
timeout 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu > $OUT/scene_tests.log 2>&1
echo "scene tests rc=$?" >> $OUT/summary.txt
tail -5 $OUT/scene_tests.log >> $OUT/summary.txt
cat $OUT/summary.txt
