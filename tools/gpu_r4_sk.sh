#!/bin/bash
# stream-K A/B session
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4_sk}
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
for sk in 1 0; do
  ES_CONV_STREAMK=$sk timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32_sk$sk.txt 2>&1
  ES_CONV_STREAMK=$sk timeout 300 python tools/conv_launch_table.py 4 > $OUT/conv_table_O4_sk$sk.txt 2>&1
  ES_CONV_STREAMK=$sk timeout 600 python tools/emulate_shards.py --steps 20 --tuned 2>&1 | grep "^world" > $OUT/shards_tuned_sk$sk.txt
  ES_CONV_STREAMK=$sk timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" > $OUT/shards_det_sk$sk.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-sub-records --reps 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -3
for sk in 1 0; do echo "== streamk $sk"; head -2 $OUT/conv_table_O32_sk$sk.txt | tail -1; head -2 $OUT/conv_table_O4_sk$sk.txt | tail -1; cat $OUT/shards_tuned_sk$sk.txt $OUT/shards_det_sk$sk.txt; done
head -c 400 $OUT/bench.json
