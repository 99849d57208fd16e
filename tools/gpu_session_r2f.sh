#!/bin/bash
# round 2, GPU session F: anti-phase hand-off variant of k_conv_ws (ES_CONV_AP=1): parity subset, then timings
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2f
ES_CONV_AP=1 ES_CONV_FORCE256=1 timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "test_conv_mfma or test_conv_fused_skip or unet3d_full_eps or test_conv_ws_at or vqvae or test_conv_down_dhw" > gpurun_out/r2f/ap_tests.log 2>&1
echo "ap tests rc=$?" >> gpurun_out/r2f/summary.txt
ES_CONV_AP=1 timeout 300 python tools/microbench_power.py > gpurun_out/r2f/power_ap.log 2>&1
timeout 300 python tools/microbench_power.py > gpurun_out/r2f/power_ws.log 2>&1
ES_CONV_AP=1 timeout 600 python bench.py --no-cpu-baseline --no-sub-records > gpurun_out/r2f/bench_ap.json 2> gpurun_out/r2f/bench_ap.err
timeout 600 python bench.py --no-cpu-baseline --no-sub-records > gpurun_out/r2f/bench_ws.json 2> gpurun_out/r2f/bench_ws.err
cat gpurun_out/r2f/summary.txt; tail -6 gpurun_out/r2f/ap_tests.log; cat gpurun_out/r2f/power_ap.log gpurun_out/r2f/power_ws.log; cut -c1-330 gpurun_out/r2f/bench_ap.json gpurun_out/r2f/bench_ws.json
