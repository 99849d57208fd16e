#!/bin/bash
# round 6, session 6: whole GPU suite on the tree with the GroupNorm part_in fix, shard emulation in both modes, the 4-objects step by kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s6}
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -n 1 > $OUT/pytest_all.txt 2>&1; tail -8 $OUT/pytest_all.txt | cut -c1-250
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" > $OUT/emu_tuned.txt
timeout 600 python tools/emulate_shards.py --steps 20 --deterministic 2>&1 | grep "^world" > $OUT/emu_exact.txt
cat $OUT/emu_tuned.txt $OUT/emu_exact.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_w8 -o w8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 20 --worlds 8 > $OUT/prof_w8.log 2>&1 )
KT=$(find $OUT/prof_w8 -name "*kernel_trace.csv" | head -1)
python tools/step_breakdown.py $KT 10 > $OUT/step_breakdown_w8.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +6M -delete
head -30 $OUT/step_breakdown_w8.txt
timeout 400 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; tail -1 $OUT/bench_quick.json | cut -c1-600
