#!/bin/bash
# round 4, late: a shard's GCN chain as a parallel branch of its main plan -- bit-exactness test, then the same-box A/B (one-GPU emulation)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-shardlane}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_vol.py -m gpu -q -x -k "shards" --durations=6 > $OUT/tests_shards.log 2>&1
echo "shard tests rc=$?" > $OUT/summary.txt
for side in 1 0; do
  ES_SHARD_SIDE=$side timeout 300 python tools/emulate_shards.py --steps 20 --worlds 8,4 --tuned 2>&1 | grep "^world" > $OUT/emu_tuned_side$side.txt
  ES_SHARD_SIDE=$side timeout 300 python tools/emulate_shards.py --steps 20 --worlds 8 2>&1 | grep "^world" > $OUT/emu_exact_side$side.txt
done
cat $OUT/summary.txt; tail -9 $OUT/tests_shards.log; head -3 $OUT/emu_*.txt
