#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2n
ES_CONV_PIPE=1 timeout 900 python tools/conv_launch_table.py > gpurun_out/r2n/table.log 2>&1
grep -v amdgpu gpurun_out/r2n/table.log | cut -c1-200
