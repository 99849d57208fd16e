#!/bin/bash
# round 5: the few-objects regime (4 objects per GPU, tuned shards): kernel stats of rank 0 of 8, emulated on one GPU
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-fewobj}
mkdir -p $OUT
python tools/emulate_shards.py --steps 20 --tuned > $OUT/emu_tuned.txt 2>&1
python tools/emulate_shards.py --steps 20 > $OUT/emu_exact.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_w8 -o w8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 20 --tuned --worlds 8 > $OUT/prof_w8.log 2>&1 )
KT=$(find $OUT/prof_w8 -name "*kernel_trace.csv" | head -1)
python tools/step_breakdown.py $KT 10 > $OUT/step_breakdown_w8.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +6M -delete
cat $OUT/emu_tuned.txt | grep world; cat $OUT/emu_exact.txt | grep world; head -40 $OUT/step_breakdown_w8.txt
