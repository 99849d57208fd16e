#!/bin/bash
# last call of a round: GPU suite + the bench line of record, taken AFTER the PMC summary it cites (roofline.traffic_source) is committed
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-line}
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py --steps 100 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
timeout 400 python tools/scene_sizes_latency.py 32,10,16 > $OUT/scene_sizes.txt 2>&1
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -2; grep -E "^FAILED|eta 0.7" $OUT/tests_gpu.log | cut -c1-200; grep -v amdgpu $OUT/scene_sizes.txt; tail -1 $OUT/bench_final.json | cut -c1-400
