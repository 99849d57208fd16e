#!/bin/bash
# round 6, session 11: k_conv_ws3b (few-objects 128-row tile with the shared A tile, in-place register shifts)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s11}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_vol.py -q -m gpu -n 1 -k "few_objects or shards or canonical or rowgroup" > $OUT/pytest_sel.txt 2>&1; tail -5 $OUT/pytest_sel.txt | cut -c1-250
ES_CONV_A3B=0 timeout 400 python tools/emulate_shards.py --steps 20 --worlds 1,2,4,8 2>&1 | grep "^world" | sed 's/^/A3B off: /'
timeout 400 python tools/emulate_shards.py --steps 20 --worlds 1,2,4,8 2>&1 | grep "^world" | sed 's/^/A3B on:  /'
ES_CONV_A3B=0 timeout 300 python tools/shard_op_table.py --world 8 2>&1 | grep -v amdgpu > $OUT/op_table_w8_a3boff.txt; head -4 $OUT/op_table_w8_a3boff.txt
timeout 300 python tools/shard_op_table.py --world 8 2>&1 | grep -v amdgpu > $OUT/op_table_w8_a3bon.txt; head -4 $OUT/op_table_w8_a3bon.txt
