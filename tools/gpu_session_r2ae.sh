#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2ae
timeout 1500 python -m pytest tests/test_hip_vol.py -m gpu -x -q > gpurun_out/r2ae/tests.log 2>&1
echo "vol tests rc=$?" > gpurun_out/r2ae/summary.txt
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" >> gpurun_out/r2ae/summary.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2ae/prof8 -o st --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 30 --worlds 8 > /dev/null 2>&1 )
python tools/step_breakdown.py $(ls gpurun_out/r2ae/prof8/*kernel_trace.csv gpurun_out/r2ae/prof8/*/*kernel_trace.csv 2>/dev/null | head -1) ddim 30 > gpurun_out/r2ae/breakdown8.txt 2>&1
find gpurun_out/r2ae -name "*kernel_trace.csv" -delete; find gpurun_out/r2ae -name "*agent_info.csv" -delete
cat gpurun_out/r2ae/summary.txt; tail -2 gpurun_out/r2ae/tests.log; grep -A14 "by kernel" gpurun_out/r2ae/breakdown8.txt; head -16 gpurun_out/r2ae/breakdown8.txt
