#!/bin/bash
# round 6, session 2: k_conv_kw (K split inside the workgroup) and deeper rings for the small producer/consumer tiles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s2}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_vol.py -x -q -m gpu -k "few_objects_kernels or test_conv_mfma or fused_skip" > $OUT/pytest_kw.txt 2>&1
tail -15 $OUT/pytest_kw.txt
timeout 500 python tools/microbench_tiles.py --O 4 > $OUT/tiles_O4.txt 2>&1
timeout 400 python tools/microbench_tiles.py --O 16 --shapes 0,2,4,6,7,9 > $OUT/tiles_O16.txt 2>&1
cat $OUT/tiles_O4.txt
