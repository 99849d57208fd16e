#!/bin/bash
# profile collection on the final tree of round 5: bench lines, kernel stats, PMC passes (each counter set in its own pass, --kernel-trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5b_prof}
mkdir -p $OUT
timeout 400 python bench.py --steps 100 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
timeout 200 python bench.py --workload layout --steps 200 --warmup 5 > $OUT/bench_layout.json 2> $OUT/bench_layout.err
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/prof_final -o st --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/prof_final.log 2>&1 )
( cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/prof_layout -o lay --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload layout --steps 50 --warmup 3 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/prof_layout.log 2>&1 )
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=$(echo $SET | tr ' ' '_' | cut -c1-24)
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_final/$D -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/pmc_$D.log 2>&1 )
done
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_layout/$SET -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload layout --steps 20 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/pmc_layout_$SET.log 2>&1 )
done
find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +6M -delete
find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; tail -1 $OUT/bench_final.json | cut -c1-300; tail -1 $OUT/bench_layout.json | cut -c1-300
