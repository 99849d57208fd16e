#!/bin/bash
# session 3: same-box A/B of the folded self-attention (NT = 2 variant), its kernel test, per-launch table, scene suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r5b_s3
mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_rows.py -x -q -m gpu -k "formed_row or unet1d_full or layout_loop_tiny or blockwise" > $OUT/rows_tests.log 2>&1
echo "rows tests rc=$?" > $OUT/summary.txt
tail -5 $OUT/rows_tests.log >> $OUT/summary.txt
timeout 400 python tools/ab_layout_fold.py 1000 5 1,0,1,0 > $OUT/ab_fold.txt 2>&1
cat $OUT/ab_fold.txt >> $OUT/summary.txt
timeout 300 python tools/layout_op_times.py > $OUT/layout_op_times.txt 2>&1
head -40 $OUT/layout_op_times.txt >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_scene.py -x -q -m gpu > $OUT/scene_tests.log 2>&1
echo "scene tests rc=$?" >> $OUT/summary.txt
tail -5 $OUT/scene_tests.log >> $OUT/summary.txt
cat $OUT/summary.txt
