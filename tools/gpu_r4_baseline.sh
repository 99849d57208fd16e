#!/bin/bash
# round-4 baseline: GPU test suite, smoke, bench line, launch tables at 32 and 4 objects, shard emulation
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4_base}
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -x -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32.txt 2>&1
timeout 300 python tools/conv_launch_table.py 4 > $OUT/conv_table_O4.txt 2>&1
timeout 300 python tools/aux_launch_table.py > $OUT/aux_table.txt 2>&1
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" > $OUT/shards_default.txt
timeout 600 python tools/emulate_shards.py --steps 20 --tuned 2>&1 | grep "^world" > $OUT/shards_tuned.txt
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -3; tail -2 $OUT/smoke.log; cat $OUT/shards_default.txt $OUT/shards_tuned.txt; head -c 600 $OUT/bench.json
