#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2o
timeout 600 python tools/microbench_ab.py --child > gpurun_out/r2o/ab.log 2>&1
timeout 1500 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py -m gpu -x -q > gpurun_out/r2o/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2o/summary.txt
timeout 900 python tools/conv_launch_table.py > gpurun_out/r2o/table.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-sub-records > gpurun_out/r2o/bench.json 2> gpurun_out/r2o/bench.err
cat gpurun_out/r2o/summary.txt; grep -v amdgpu gpurun_out/r2o/ab.log; tail -3 gpurun_out/r2o/tests.log; cut -c1-300 gpurun_out/r2o/bench.json
