#!/bin/bash
# round 6, session 1: where the few-objects step spends its time (per-op tables at 4 objects per GPU, both shard modes) and the
# 64- / 128-row producer/consumer tiles against the dispatcher's choice
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s1}
mkdir -p $OUT
timeout 300 python tools/shard_op_table.py --world 8 --tuned > $OUT/op_table_w8_tuned.txt 2>&1
timeout 300 python tools/shard_op_table.py --world 8 > $OUT/op_table_w8_exact.txt 2>&1
timeout 400 python tools/microbench_tiles.py --O 4 > $OUT/tiles_O4.txt 2>&1
timeout 400 python tools/microbench_tiles.py --O 16 --shapes 0,2,4,6,7,9 > $OUT/tiles_O16.txt 2>&1
timeout 600 python -m pytest tests/test_hip_rows.py -x -q -m gpu > $OUT/pytest_rows.txt 2>&1
tail -3 $OUT/pytest_rows.txt
head -12 $OUT/op_table_w8_tuned.txt; head -12 $OUT/op_table_w8_exact.txt
cat $OUT/tiles_O4.txt
