cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_scene.py tests/test_hip_vol.py -x -q -k "vqvae or groupnorm or gn or unet3d_full_eps_vs" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/vq -o vq --output-format csv -- python $GRAFT_REPO_ROOT/tools/profile_vq.py > $OUT/vq.log 2>&1 )
grep "decode of" $OUT/vq.log
python - <<'PY'
import csv, glob, re
f = (glob.glob('/root/repo/gpurun_out/r3j/vq/*kernel_stats.csv') + glob.glob('/root/repo/gpurun_out/r3j/vq/*/*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:10]:
    n = re.search(r'k_[a-z0-9_]+(<[^>]*>)?', r['Name']); n = n.group(0) if n else r['Name'][:40]
    print('%-40s calls %5s avg %9.1f us  %5.1f%%' % (n, r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
