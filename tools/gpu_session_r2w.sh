#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2w
for rep in 1 2; do for v in 0 1 2 3; do echo "ES_ATT_VARIANT=$v" >> gpurun_out/r2w/att.log; ES_ATT_VARIANT=$v timeout 300 python tools/microbench_attention.py 2>&1 | grep -v amdgpu >> gpurun_out/r2w/att.log; done; done
timeout 600 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "attention or unet3d_full_eps or vqvae" > gpurun_out/r2w/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2w/att.log
cat gpurun_out/r2w/att.log; tail -2 gpurun_out/r2w/tests.log
