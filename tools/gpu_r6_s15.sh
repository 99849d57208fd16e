#!/bin/bash
# round 6, session 15: counters of the 4-objects step (rank 0 of 8 emulated): MFMA busy and L2 fetch per kernel family
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s15}
mkdir -p $OUT
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$(echo $SET | tr ' ' '_' | cut -c1-24)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_w8/$D -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 10 --worlds 8 > $OUT/pmc_$D.log 2>&1 )
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete
python - <<'PY'
import csv, glob, collections, re, os
out=os.environ.get('OUTD', '')
PY
ls -R $OUT | head -30; du -sh $OUT
