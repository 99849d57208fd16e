#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2ab
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep -v amdgpu > gpurun_out/r2ab/shards.log
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2ab/prof8 -o st --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 30 --worlds 8 > /dev/null 2>&1 )
python tools/step_breakdown.py $(ls gpurun_out/r2ab/prof8/*kernel_trace.csv gpurun_out/r2ab/prof8/*/*kernel_trace.csv 2>/dev/null | head -1) ddim 45 > gpurun_out/r2ab/breakdown8.txt 2>&1
find gpurun_out/r2ab -name "*kernel_trace.csv" -delete; find gpurun_out/r2ab -name "*agent_info.csv" -delete
tail -6 gpurun_out/r2ab/shards.log; tail -30 gpurun_out/r2ab/breakdown8.txt
