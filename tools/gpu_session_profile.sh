#!/bin/bash
# final profile collection of a round: bench line, kernel stats, PMC passes (each counter set in its own pass)
# usage: bash tools/gpu_session_profile.sh <out-subdir>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
python bench.py --steps 100 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
python bench.py --workload layout --steps 200 --warmup 5 > $OUT/bench_layout.json 2> $OUT/bench_layout.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_final -o st --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/prof_final.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_configs4 -o c4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --scaling weak --scenes-per-gpu 8 --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/prof_configs4.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_layout -o lay --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload layout --steps 50 --warmup 3 --reps 1 --no-cpu-baseline > $OUT/prof_layout.log 2>&1 )
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  D=$(echo $SET | tr ' ' '_' | cut -c1-24)
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_final/$D -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-records > $OUT/pmc_$D.log 2>&1 )
done
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_layout/$SET -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload layout --steps 20 --warmup 2 --reps 1 --no-cpu-baseline > $OUT/pmc_layout_$SET.log 2>&1 )
done
timeout 600 python tools/microbench_rows.py cold 2>&1 | grep "^cold" > $OUT/rows_microbench.txt
timeout 600 python tools/conv_launch_table.py 2>&1 | grep -v amdgpu > $OUT/conv_launch_table.txt
timeout 600 python tools/aux_launch_table.py 2>&1 | grep -v amdgpu > $OUT/aux_launch_table.txt
timeout 300 python tools/layout_op_times.py 2>&1 | grep -v amdgpu > $OUT/layout_op_times.txt
# keep the merge small: the kernel traces of the PMC passes are not needed
find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +6M -delete
find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; tail -2 $OUT/bench_final.json | cut -c1-300
