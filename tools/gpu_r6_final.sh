#!/bin/bash
# final GPU session of round 6, in order of importance, every step under its own timeout: the bench line of record, the layout line,
# rocprofv3 kernel stats of both, PMC passes (each counter set in its own pass, --kernel-trace only), per-launch tables, shard
# emulation in both modes, scene-call latency, then the whole GPU suite and smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_final}
mkdir -p $OUT
timeout 500 python bench.py --steps 100 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
echo "bench rc=$?" > $OUT/summary.txt
timeout 300 python bench.py --workload layout --steps 200 --warmup 5 > $OUT/bench_layout.json 2> $OUT/bench_layout.err
echo "bench layout rc=$?" >> $OUT/summary.txt
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_final -o st --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/prof_final.log 2>&1 )
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/prof_layout -o lay --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload layout --steps 50 --warmup 3 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/prof_layout.log 2>&1 )
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=$(echo $SET | tr ' ' '_' | cut -c1-24)
  ( cd /tmp && timeout 250 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_final/$D -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/pmc_$D.log 2>&1 )
done
find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
timeout 300 python tools/conv_launch_table.py 32 2>&1 | grep -v amdgpu > $OUT/conv_launch_table.txt
timeout 300 python tools/aux_launch_table.py 2>&1 | grep -v amdgpu > $OUT/aux_launch_table.txt
timeout 200 python tools/layout_op_times.py 2>&1 | grep -v amdgpu > $OUT/layout_op_times.txt
timeout 400 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" > $OUT/shards_tuned.txt
timeout 400 python tools/emulate_shards.py --steps 20 --deterministic 2>&1 | grep "^world" > $OUT/shards_exact.txt
timeout 300 python tools/shard_op_table.py --world 8 2>&1 | grep -v amdgpu > $OUT/op_table_w8.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_w8 -o w8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --steps 20 --worlds 8 > $OUT/prof_w8.log 2>&1 )
KT=$(find $OUT/prof_w8 -name "*kernel_trace.csv" | head -1)
python tools/step_breakdown.py $KT 10 > $OUT/step_breakdown_w8.txt 2>&1
timeout 200 python tools/e2e_latency.py > $OUT/e2e.txt 2>&1
timeout 200 python tools/scene_sizes_latency.py > $OUT/scene_sizes.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +6M -delete
find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete
timeout 2400 python -m pytest tests -m gpu -q -n 1 > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" >> $OUT/summary.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -3; tail -2 $OUT/smoke.log; tail -1 $OUT/bench_final.json | cut -c1-400; tail -1 $OUT/bench_layout.json | cut -c1-300; grep -v amdgpu $OUT/e2e.txt | tail -6; cat $OUT/shards_tuned.txt $OUT/shards_exact.txt; du -sh $OUT
