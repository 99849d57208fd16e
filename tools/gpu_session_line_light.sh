#!/bin/bash
# the bench line of record, taken AFTER the PMC summary it cites (roofline.traffic_source) is committed; smoke first
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-line}
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" > $OUT/summary.txt
timeout 900 python bench.py --steps 100 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
cat $OUT/summary.txt; tail -1 $OUT/bench_final.json | cut -c1-600
