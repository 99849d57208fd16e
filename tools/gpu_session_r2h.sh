#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_mc.py -m gpu -x -q > gpurun_out/r2h/mc.log 2>&1
echo "mc rc=$?" >> gpurun_out/r2h/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-sub-records --fuse-loops 1 > gpurun_out/r2h/bench_fused.json 2> gpurun_out/r2h/bench_fused.err
timeout 600 python bench.py --no-cpu-baseline --no-sub-records --fuse-loops 0 > gpurun_out/r2h/bench_two.json 2> gpurun_out/r2h/bench_two.err
timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "test_conv or unet3d_full_eps or shards" > gpurun_out/r2h/vol.log 2>&1
echo "vol rc=$?" >> gpurun_out/r2h/summary.txt
cat gpurun_out/r2h/summary.txt; tail -4 gpurun_out/r2h/mc.log; cut -c1-900 gpurun_out/r2h/bench_fused.json; tail -3 gpurun_out/r2h/bench_fused.err; cut -c1-330 gpurun_out/r2h/bench_two.json; tail -3 gpurun_out/r2h/vol.log
