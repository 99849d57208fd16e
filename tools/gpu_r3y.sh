cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "conv_mfma or fused_skip or ws_at or rowgroup or full_eps or down_dhw" 2>&1 | tail -5
timeout 600 python tools/microbench_linear.py 2>&1 | tail -8
timeout 600 python tools/conv_launch_table.py 2>&1 | head -30
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sub-records 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('full %.3f ms  shape %.3f ms  frac %.4f  value %.3f' % (d['ms_per_step'], d['config']['shape']['ms_per_step'], d['roofline']['frac'], d['value']))
"
