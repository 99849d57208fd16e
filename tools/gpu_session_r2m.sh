#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2m
timeout 900 python tools/microbench_ab.py ES_CONV_PIPE > gpurun_out/r2m/ab.log 2>&1
for v in 0 1 0 1; do ES_CONV_PIPE=$v timeout 600 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('PIPE=$v', d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" >> gpurun_out/r2m/bench.log; done
ES_CONV_PIPE=1 ES_CONV_FORCE256=1 timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "test_conv_mfma or test_conv_fused_skip or unet3d_full_eps or test_conv_ws_at or test_conv_down_dhw or bitwise" > gpurun_out/r2m/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2m/bench.log
grep -v amdgpu gpurun_out/r2m/ab.log; cat gpurun_out/r2m/bench.log; tail -3 gpurun_out/r2m/tests.log
