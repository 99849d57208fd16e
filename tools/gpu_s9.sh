#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s9
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_rows.py tests/test_hip_vol.py -m gpu -x -q -k "fold_product or from_a_model or splitk_reduction" 2>&1 | tail -5
python tools/e2e_latency.py --profile-first > $OUT/e2e_first.txt 2>&1
grep -E "^call|loop|decode|GCN|new graph" $OUT/e2e_first.txt
python tools/e2e_latency.py --prewarm 12 > $OUT/e2e_prewarm.txt 2>&1
grep -E "^call|prewarm" $OUT/e2e_prewarm.txt
python -c "import torch; print(torch.cuda.memory_reserved())" 
