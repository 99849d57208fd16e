#!/bin/bash
# last session of the round: the full GPU suite and smoke on the final tree, then the kernel breakdown of the 4-objects tuned step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5b_verify}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
tail -3 $OUT/tests_gpu.log >> $OUT/summary.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/summary.txt
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_w8 -o w8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 20 --tuned --worlds 8 > $OUT/prof_w8.log 2>&1 )
KT=$(find $OUT/prof_w8 -name "*kernel_trace.csv" | head -1)
timeout 100 python tools/step_breakdown.py $KT 10 > $OUT/step_breakdown_w8.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary.txt; head -16 $OUT/step_breakdown_w8.txt
