#!/bin/bash
# two column tiles per workgroup for the GEGLU projection of the layout denoiser: tests of everything that runs it, then the same-box A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-nt2}
mkdir -p $OUT
timeout 150 python -m pytest tests/test_hip_rows.py -m gpu -q -x > $OUT/tests_rows.log 2>&1
echo "rows tests rc=$?" > $OUT/summary.txt
timeout 60 python -m pytest tests/test_hip_traj.py -m gpu -q -x -k layout > $OUT/tests_traj_layout.log 2>&1
echo "traj layout rc=$?" >> $OUT/summary.txt
for v in 1 0 1 0; do
  ES_ROWS_NT2=$v timeout 60 python tools/ab_layout_ride.py 1000 5 2,2 2>&1 | grep -v amdgpu >> $OUT/ab_nt2.txt
done
timeout 100 python -m pytest tests/test_hip_scene.py -m gpu -q -x -k "sgdiff_api or model_files" > $OUT/tests_scene.log 2>&1
echo "scene subset rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -2 $OUT/tests_rows.log; tail -2 $OUT/tests_traj_layout.log; tail -2 $OUT/tests_scene.log; cat $OUT/ab_nt2.txt
