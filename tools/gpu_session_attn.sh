#!/bin/bash
# attention: masking behind a branch + double-buffered K / V tiles -- correctness, per-launch table, step A/B
tag=${1:-attn}
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "attention or full_eps_O32 or vqvae or concat" > $out/tests.log 2>&1
echo "tests rc=$?" > $out/summary.txt
tail -3 $out/tests.log
timeout 300 python tools/aux_launch_table.py > $out/aux_default.txt 2>&1
ES_ATTN_NBUF=1 timeout 300 python tools/aux_launch_table.py > $out/aux_nbuf1.txt 2>&1
for f in default nbuf1; do echo "== $f"; head -1 $out/aux_$f.txt; grep "attn" $out/aux_$f.txt; done
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --no-sub-records --reps 3 > $out/bench_default_$i.json 2>$out/bench.err
ES_ATTN_NBUF=1 timeout 400 python bench.py --no-cpu-baseline --no-sub-records --reps 3 > $out/bench_nbuf1_$i.json 2>>$out/bench.err
done
for f in $out/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', d['value'], d['repetitions']['shape_ms_per_step']['median'])
"; done
cat $out/summary.txt
