cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $OUT
timeout 600 python tools/probe_step_graph.py 2>&1 | grep -v amdgpu | tail -5
timeout 900 python tools/probe_step_graph.py 224 2>&1 | grep -v amdgpu | tail -3
timeout 900 python bench.py --steps 60 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r3f/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('traffic_source'))
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:900])
for r in d.get('sub_records', []):
    print(json.dumps(r)[:700])
PY
tail -3 $OUT/bench.err
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" | tee $OUT/shards_default.txt
timeout 600 python tools/emulate_shards.py --steps 20 --deterministic 2>&1 | grep "^world" | tee $OUT/shards_det.txt
