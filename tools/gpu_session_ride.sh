#!/bin/bash
# round 4, late: the head of the layout trunk riding on the GCN launches -- rows tests, then the same-box A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-ride}
mkdir -p $OUT
timeout 500 python -m pytest tests/test_hip_rows.py -m gpu -q -x --durations=8 > $OUT/tests_rows.log 2>&1
echo "rows tests rc=$?" > $OUT/summary.txt
timeout 200 python -m pytest tests/test_hip_traj.py -m gpu -q -x -k layout > $OUT/tests_traj_layout.log 2>&1
echo "traj layout rc=$?" >> $OUT/summary.txt
for u in 1 0 2; do
  ES_ROWS_U1=$u timeout 200 python tools/ab_layout_ride.py 1000 5 2>&1 | grep -v amdgpu > $OUT/ab_u$u.txt
done
cat $OUT/summary.txt; tail -4 $OUT/tests_rows.log; tail -2 $OUT/tests_traj_layout.log; cat $OUT/ab_u1.txt $OUT/ab_u0.txt $OUT/ab_u2.txt
