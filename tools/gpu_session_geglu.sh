#!/bin/bash
# GEGLU epilogue: v_rcp instead of the IEEE quotient, bank-masked DPP exchange -- correctness, per-launch table, step
tag=${1:-geglu}
out=gpurun_out/$tag
mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "geglu or gelu or attention or shards_equal or full_eps or deep or alternate" > $out/tests.log 2>&1
echo "tests rc=$?" > $out/summary.txt
tail -3 $out/tests.log
timeout 300 python tools/conv_launch_table.py > $out/table.txt 2>&1
head -1 $out/table.txt; grep "epi 1" $out/table.txt
timeout 300 python tools/aux_launch_table.py > $out/aux.txt 2>&1
head -1 $out/aux.txt; grep attn $out/aux.txt
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --no-sub-records --reps 3 > $out/bench_$i.json 2>$out/bench.err
done
for f in $out/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', d['value'], d['repetitions']['shape_ms_per_step']['median'])
"; done
cat $out/summary.txt
