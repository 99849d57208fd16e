cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_vol.py tests/test_hip_rows.py -x -q -k "stem or unet3d_tiny or unet3d_full_eps_vs or shards_equal or multi_problem" 2>&1 | tail -3
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/sh8 -o sh8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 20 --worlds 1,8 > $OUT/sh8.log 2>&1 )
grep "^world" $OUT/sh8.log
python - <<'PY'
import csv, glob, re
f = (glob.glob('/root/repo/gpurun_out/r3m/sh8/*kernel_stats.csv') + glob.glob('/root/repo/gpurun_out/r3m/sh8/*/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    if 'stem' in r['Name']:
        print(r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, 'us')
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
