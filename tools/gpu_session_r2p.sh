#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p
timeout 1200 python tools/microbench_ab.py ES_CONV_WS128 > gpurun_out/r2p/ab.log 2>&1
ES_CONV_WS128=1 ES_CONV_FORCE256=1 timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "test_conv_mfma or test_conv_fused_skip or unet3d_full_eps or test_conv_ws_at or test_conv_down_dhw or geglu" > gpurun_out/r2p/tests.log 2>&1
echo "tests128 rc=$?" >> gpurun_out/r2p/bench.log
for v in 0 1; do ES_CONV_WS128=$v timeout 600 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('WS128=$v', d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])" >> gpurun_out/r2p/bench.log; done
grep -v amdgpu gpurun_out/r2p/ab.log; cat gpurun_out/r2p/bench.log; tail -3 gpurun_out/r2p/tests.log
