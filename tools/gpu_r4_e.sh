#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4_e}
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-sub-records --reps 3"
for t in 256 512 768; do
  ES_CONV_WSS_TARGET=$t timeout 400 $B > $OUT/ab_t$t.json 2>/dev/null
  ES_CONV_WSS_TARGET=$t timeout 600 python tools/emulate_shards.py --steps 20 --worlds 1,8 2>&1 | grep "^world" > $OUT/shards_det_t$t.txt
  ES_CONV_WSS_TARGET=$t timeout 600 python tools/emulate_shards.py --steps 20 --worlds 1,2,4,8 --tuned 2>&1 | grep "^world" > $OUT/shards_tuned_t$t.txt
done
timeout 400 $B > $OUT/ab_t256b.json 2>/dev/null
for t in 256 512 768 256b; do python - <<PY
import json
try:
    d=json.load(open('$OUT/ab_t$t.json')); print('$t', d['value'], d['value_min_max'], 'shape', d['config']['shape']['ms_per_step'])
except Exception as e: print('$t', 'ERR', e)
PY
done
for t in 256 512 768; do echo "== $t"; cat $OUT/shards_det_t$t.txt $OUT/shards_tuned_t$t.txt; done
