#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s8
mkdir -p $OUT
python tools/e2e_latency.py --profile-first > $OUT/e2e_first.txt 2>&1
grep -E "call|loop|decode|GCN" $OUT/e2e_first.txt
timeout 600 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "splitk_reduction" 2>&1 | tail -3
