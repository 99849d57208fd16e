#!/bin/bash
# round 6, session 8: fp32 attention on the matrix instruction (fp32 / fp32x routes), refined few-objects rules
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s8}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py -q -m gpu -n 1 -k "fp32_operand or few_objects or shards or canonical or deep_ring" > $OUT/pytest_sel.txt 2>&1; tail -6 $OUT/pytest_sel.txt | cut -c1-250
timeout 300 python tools/profile_fp32x.py 32 fp32x 2>&1 | tail -1
timeout 300 python tools/profile_fp32x.py 32 fp32 2>&1 | tail -1
ES_ATTN_F32_SCALAR=1 timeout 300 python tools/profile_fp32x.py 32 fp32x 2>&1 | tail -1
timeout 400 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world"
timeout 300 python tools/conv_launch_table.py 16 2>&1 | grep -v amdgpu > $OUT/conv_table_O16.txt; head -1 $OUT/conv_table_O16.txt
timeout 300 python tools/shard_op_table.py --world 8 2>&1 | grep -v amdgpu > $OUT/op_table_w8.txt; head -8 $OUT/op_table_w8.txt
