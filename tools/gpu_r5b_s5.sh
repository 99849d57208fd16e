#!/bin/bash
# one-round row statistics in the formed-row launch: parity tests, then the same-box A/B against the unfolded plan
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5b_s5}
mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_rows.py -x -q -m gpu -k "formed_row or unet1d or layout_loop or blockwise" > $OUT/rows_tests.log 2>&1
echo "rows tests rc=$?" > $OUT/summary.txt; tail -3 $OUT/rows_tests.log >> $OUT/summary.txt
timeout 200 python -m pytest tests/test_hip_traj.py -x -q -m gpu -k layout > $OUT/traj.log 2>&1
echo "traj rc=$?" >> $OUT/summary.txt; tail -2 $OUT/traj.log >> $OUT/summary.txt
timeout 300 python tools/ab_layout_fold.py 1000 5 1,0,1,0 > $OUT/ab_fold.txt 2>&1
grep "^fold" $OUT/ab_fold.txt >> $OUT/summary.txt
cat $OUT/summary.txt
