#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2v
timeout 1200 python tools/microbench_ab.py ES_CONV_WS128 > gpurun_out/r2v/ab.log 2>&1
for v in 0 1 0 1; do ES_CONV_WS128=$v timeout 600 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('WS128=$v', d['value'], d['config']['shape']['ms_per_step'])" >> gpurun_out/r2v/bench.log; done
ES_CONV_WS128=1 timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "unet3d_full_eps or bitwise or test_conv_mfma or ddim_tiny" > gpurun_out/r2v/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2v/bench.log
grep -v amdgpu gpurun_out/r2v/ab.log | sed 's/|/\n   /g' | grep "ES_CONV\|16x4x4"; cat gpurun_out/r2v/bench.log; tail -2 gpurun_out/r2v/tests.log
