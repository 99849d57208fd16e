#!/bin/bash
# round 2, GPU session D: k_conv_ws two units per barrier, deterministic sharding, layout fold, marching cubes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
python tools/microbench_power.py > gpurun_out/r2d/power_ups2.log 2>&1
ES_CONV_UPS=1 python tools/microbench_power.py > gpurun_out/r2d/power_ups1.log 2>&1
python -m pytest tests/test_mc.py -m gpu -x -q > gpurun_out/r2d/mc.log 2>&1
echo "mc rc=$?" >> gpurun_out/r2d/summary.txt
timeout 1800 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py tests/test_hip_rows.py -m gpu -x -q > gpurun_out/r2d/vol.log 2>&1
echo "vol rc=$?" >> gpurun_out/r2d/summary.txt
python bench.py --no-cpu-baseline > gpurun_out/r2d/bench_full.json 2> gpurun_out/r2d/bench_full.err
ES_CONV_UPS=1 python bench.py --no-cpu-baseline > gpurun_out/r2d/bench_full_ups1.json 2> gpurun_out/r2d/bench_full_ups1.err
python bench.py --workload layout --no-cpu-baseline > gpurun_out/r2d/bench_layout.json 2> gpurun_out/r2d/bench_layout.err
timeout 900 python -m pytest tests/test_hip_scene.py -m gpu -x -q > gpurun_out/r2d/scene.log 2>&1
echo "scene rc=$?" >> gpurun_out/r2d/summary.txt
cat gpurun_out/r2d/summary.txt; tail -5 gpurun_out/r2d/mc.log; tail -8 gpurun_out/r2d/vol.log; cat gpurun_out/r2d/power_ups2.log gpurun_out/r2d/power_ups1.log; cut -c1-330 gpurun_out/r2d/bench_full.json gpurun_out/r2d/bench_full_ups1.json gpurun_out/r2d/bench_layout.json; tail -3 gpurun_out/r2d/scene.log
