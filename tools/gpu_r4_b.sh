#!/bin/bash
# round-4 session B: output-conv kernel + f16-only ResBlock intermediates
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4_b}
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32.txt 2>&1
timeout 300 python tools/aux_launch_table.py > $OUT/aux_table.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --reps 3 > $OUT/bench.json 2> $OUT/bench.err
ES_GN_F16=0 timeout 600 python bench.py --no-cpu-baseline --no-sub-records --reps 3 > $OUT/bench_nof16.json 2> $OUT/bench_nof16.err
ES_CONV_N16=0 timeout 600 python bench.py --no-cpu-baseline --no-sub-records --reps 3 > $OUT/bench_non16.json 2> $OUT/bench_non16.err
timeout 300 python tools/profile_vq.py > $OUT/vq.txt 2>&1
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -3; head -2 $OUT/conv_table_O32.txt | tail -1; grep "N    3" $OUT/conv_table_O32.txt
for f in bench bench_nof16 bench_non16; do python -c "
import json,sys; d=json.load(open('$OUT/$f.json')); print('$f', d['value'], d['value_min_max'], d['config']['shape']['ms_per_step'], d['roofline']['achieved'])"; done
tail -3 $OUT/vq.txt
