#!/bin/bash
# round 6, session 16: phase stamps of the few-objects 3x3x3 launch (128-row producer/consumer tiles) -- the ~20 us outside the K loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s16}
mkdir -p $OUT
ES_LIB_TAG=_stamp timeout 300 python tools/conv_stamps_few.py 4 2>&1 | grep -v amdgpu > $OUT/stamps_O4.txt
cat $OUT/stamps_O4.txt
