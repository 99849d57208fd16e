#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2u
timeout 600 python -m pytest tests/test_hip_vol.py -m gpu -x -q  > gpurun_out/r2u/tests0.log 2>&1
echo "tests0 rc=$?" > gpurun_out/r2u/summary.txt
timeout 1500 python -m pytest tests/test_hip_traj.py tests/test_hip_scene.py -m gpu -x -q > gpurun_out/r2u/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2u/summary.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2u/prof -o st --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sub-records > $GRAFT_REPO_ROOT/gpurun_out/r2u/prof.log 2>&1 )
find gpurun_out/r2u -name "*kernel_trace.csv" -delete; find gpurun_out/r2u -name "*agent_info.csv" -delete
timeout 600 python bench.py --no-cpu-baseline --no-sub-records > gpurun_out/r2u/bench.json 2> gpurun_out/r2u/bench.err
cat gpurun_out/r2u/summary.txt; tail -3 gpurun_out/r2u/tests0.log; tail -3 gpurun_out/r2u/tests.log; cut -c1-300 gpurun_out/r2u/bench.json
head -12 gpurun_out/r2u/prof/*/*kernel_stats.csv 2>/dev/null | cut -c1-160 || head -12 gpurun_out/r2u/prof/*kernel_stats.csv | cut -c1-160
