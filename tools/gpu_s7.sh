#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s7
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "splitk_reduction or shards or golden or rowgroup" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
bash tools/gpu_fewobj.sh s7
