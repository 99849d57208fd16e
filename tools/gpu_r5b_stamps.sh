#!/bin/bash
# where the formed-row (ES_PRO_LN_ATTN) launch spends its prologue: three instrumented builds, stamp 3 after the LayerNorm (0), after the
# statistics of t0 (1), after the row is formed (2).  Build them in the container first (they travel with the snapshot):
#   for v in 0 1 2; do ES_BUILD_TAG=_stamp$v ES_BUILD_FLAGS="-DES_STAMP -DES_STAMP_P3=$v" python -m echoscene_amd.build --force; done
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5b_stamps}
mkdir -p $OUT
for v in 0 1 2; do
  ES_LIB_TAG=_stamp$v timeout 200 python tools/rows_stamps.py 32 > $OUT/stamps_p3_$v.txt 2>&1
done
grep -E " ln|^# sums|^# step" $OUT/stamps_p3_0.txt | head -8
grep -E " ln" $OUT/stamps_p3_1.txt | head -3
grep -E " ln" $OUT/stamps_p3_2.txt | head -3
