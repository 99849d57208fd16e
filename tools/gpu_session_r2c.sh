#!/bin/bash
# round 2, GPU session C: k_conv_ws3 (A tile shared over kw) + marching cubes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
python -m pytest tests/test_mc.py -m gpu -x -q > gpurun_out/r2c/mc.log 2>&1
echo "mc rc=$?" >> gpurun_out/r2c/summary.txt
timeout 1500 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py -m gpu -x -q > gpurun_out/r2c/vol.log 2>&1
echo "vol rc=$?" >> gpurun_out/r2c/summary.txt
python tools/microbench_power.py > gpurun_out/r2c/power_kw3.log 2>&1
ES_CONV_KW3=0 python tools/microbench_power.py > gpurun_out/r2c/power_ws.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r2c/bench_full.json 2> gpurun_out/r2c/bench_full.err
ES_CONV_KW3=0 python bench.py --no-cpu-baseline > gpurun_out/r2c/bench_full_ws.json 2> gpurun_out/r2c/bench_full_ws.err
timeout 900 python -m pytest tests/test_hip_scene.py -m gpu -x -q > gpurun_out/r2c/scene.log 2>&1
echo "scene rc=$?" >> gpurun_out/r2c/summary.txt
cat gpurun_out/r2c/summary.txt; tail -15 gpurun_out/r2c/mc.log; tail -15 gpurun_out/r2c/vol.log; cat gpurun_out/r2c/power_kw3.log gpurun_out/r2c/power_ws.log; cut -c1-400 gpurun_out/r2c/bench_full.json gpurun_out/r2c/bench_full_ws.json
