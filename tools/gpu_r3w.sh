cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sub-records 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('full %.3f ms  shape %.3f ms  frac %.4f  value %.3f' % (d['ms_per_step'], d['config']['shape']['ms_per_step'], d['roofline']['frac'], d['value']))
"
