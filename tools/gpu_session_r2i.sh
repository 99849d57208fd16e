#!/bin/bash
# round 2, GPU session I: full GPU suite after the k_conv_mfma retirement + fused loops in the API; e2e latency; strong-scaling emulation
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2i
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r2i/all.log 2>&1
echo "all rc=$?" >> gpurun_out/r2i/summary.txt
timeout 600 python tools/e2e_latency.py > gpurun_out/r2i/e2e.log 2>&1
timeout 900 python tools/emulate_shards.py --steps 20 > gpurun_out/r2i/shards.log 2>&1
cat gpurun_out/r2i/summary.txt; tail -8 gpurun_out/r2i/all.log; grep -v amdgpu gpurun_out/r2i/e2e.log; grep -v amdgpu gpurun_out/r2i/shards.log
