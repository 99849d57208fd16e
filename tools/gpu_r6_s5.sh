#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s5}
mkdir -p $OUT
ES_DEBUG_SYNC=1 timeout 300 python tools/dbg_nomp_shards.py 2 0,1 > $OUT/dbg_w2.txt 2>&1; tail -4 $OUT/dbg_w2.txt | cut -c1-300
ES_DEBUG_SYNC=1 timeout 300 python tools/dbg_nomp_shards.py 1 0 > $OUT/dbg_w1.txt 2>&1; tail -2 $OUT/dbg_w1.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py -q -m gpu -n 1 -k "fp32_operand or deep_ring or few_objects" > $OUT/pytest_sel.txt 2>&1; tail -15 $OUT/pytest_sel.txt | cut -c1-250
