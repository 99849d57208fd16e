#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2l
timeout 300 python tools/microbench_power.py > gpurun_out/r2l/power.log 2>&1
timeout 300 python tools/microbench_linear.py > gpurun_out/r2l/linear.log 2>&1
timeout 1500 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py -m gpu -x -q > gpurun_out/r2l/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2l/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-sub-records > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err
cat gpurun_out/r2l/summary.txt; grep -v amdgpu gpurun_out/r2l/power.log; grep -v amdgpu gpurun_out/r2l/linear.log; tail -3 gpurun_out/r2l/tests.log; cut -c1-300 gpurun_out/r2l/bench.json
