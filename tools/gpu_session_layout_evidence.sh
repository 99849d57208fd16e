#!/bin/bash
# per-op cost of the layout step in context + HBM traffic of the rows kernels (PMC, one counter per pass) after the riding change
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-layev}
mkdir -p $OUT
timeout 200 python tools/layout_op_times.py 2>&1 | grep -v amdgpu > $OUT/layout_op_times.txt
for SET in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc_layout/$SET -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload layout --steps 20 --warmup 2 --reps 1 --no-cpu-baseline > $OUT/pmc_layout_$SET.log 2>&1 )
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete
head -12 $OUT/layout_op_times.txt; du -sh $OUT
