#!/bin/bash
# the bench lines of record of the second half of round 5 (default command lines, timed), then a planner-constant A/B under the fold
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5b_line}
mkdir -p $OUT
S=$(date +%s)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench (no flags) rc=$? wall $(( $(date +%s) - S )) s" > $OUT/summary.txt
S=$(date +%s)
timeout 600 python bench.py --steps 100 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
echo "bench --steps 100 rc=$? wall $(( $(date +%s) - S )) s" >> $OUT/summary.txt
S=$(date +%s)
timeout 400 python bench.py --workload layout --steps 200 --warmup 5 > $OUT/bench_layout.json 2> $OUT/bench_layout.err
echo "bench layout rc=$? wall $(( $(date +%s) - S )) s" >> $OUT/summary.txt
timeout 300 python tools/ab_layout_fold.py 1000 5 1,1:ROWS_LN_SPLIT=1,0,1,1:ROWS_LN_SPLIT=1 > $OUT/ab_fold_lnsplit.txt 2>&1
grep "^fold" $OUT/ab_fold_lnsplit.txt >> $OUT/summary.txt
cat $OUT/summary.txt; tail -1 $OUT/bench_final.json | cut -c1-600; tail -1 $OUT/bench_layout.json | cut -c1-400
