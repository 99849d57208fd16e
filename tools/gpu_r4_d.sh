#!/bin/bash
# round-4 session D: full GPU suite + same-box A/B of the round's changes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4_d}
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
B="python bench.py --no-cpu-baseline --no-sub-records --reps 3"
timeout 400 $B > $OUT/ab_default.json 2> $OUT/ab_default.err
ES_GN_F16=0 timeout 400 $B > $OUT/ab_nof16.json 2>/dev/null
ES_CONV_N16=0 timeout 400 $B > $OUT/ab_non16.json 2>/dev/null
ES_VOL_FOLD_FFO=0 timeout 400 $B > $OUT/ab_noffo.json 2>/dev/null
ES_LIN_RING=3 timeout 400 $B > $OUT/ab_ring3.json 2>/dev/null
ES_GN_F16=0 ES_CONV_N16=0 ES_VOL_FOLD_FFO=0 ES_LIN_RING=3 timeout 400 $B > $OUT/ab_alloff.json 2>/dev/null
timeout 400 $B > $OUT/ab_default2.json 2>/dev/null
timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32.txt 2>&1
timeout 300 python tools/aux_launch_table.py > $OUT/aux_table.txt 2>&1
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" > $OUT/shards_det.txt
timeout 600 python tools/emulate_shards.py --steps 20 --tuned 2>&1 | grep "^world" > $OUT/shards_tuned.txt
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -2; grep -E "^FAILED|fp32-operand|fp16-operand route" $OUT/tests_gpu.log | cut -c1-220
for f in default nof16 non16 noffo ring3 alloff default2; do python - <<PY
import json
try:
    d=json.load(open('$OUT/ab_$f.json')); print('$f', d['value'], d['value_min_max'], 'shape', d['config']['shape']['ms_per_step'], d['config']['shape']['kernels_per_step'], 'roof', d['roofline']['achieved'])
except Exception as e: print('$f', 'ERR', e)
PY
done
cat $OUT/shards_det.txt $OUT/shards_tuned.txt
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
for r in d['sub_records']: print(str(r.get('config'))[:60], r.get('value'), r.get('ms_per_step'), r.get('roofline',{}).get('frac') if r.get('roofline') else None, r.get('fp16_product_vs_fp32_route_eps'), r.get('error'))
PY
