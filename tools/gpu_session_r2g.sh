#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2g
ES_CONV_AP=2 timeout 300 python tools/microbench_power.py > gpurun_out/r2g/power_prio.log 2>&1
timeout 300 python tools/microbench_power.py > gpurun_out/r2g/power_ws.log 2>&1
ES_CONV_AP=2 timeout 600 python bench.py --no-cpu-baseline --no-sub-records > gpurun_out/r2g/bench_prio.json 2> gpurun_out/r2g/bench_prio.err
timeout 600 python bench.py --no-cpu-baseline --no-sub-records > gpurun_out/r2g/bench_ws.json 2> gpurun_out/r2g/bench_ws.err
# where do the waves wait?  SQ buckets for the conv microbench (all waves of the kernel: 8 consumers + 4 producers)
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM -d $GRAFT_REPO_ROOT/gpurun_out/r2g/pmc_sq -o sq --output-format csv -- python $GRAFT_REPO_ROOT/tools/microbench_power.py > $GRAFT_REPO_ROOT/gpurun_out/r2g/pmc_sq.log 2>&1 )
find gpurun_out/r2g -name "*agent_info.csv" -delete
cat gpurun_out/r2g/power_prio.log gpurun_out/r2g/power_ws.log; cut -c1-330 gpurun_out/r2g/bench_prio.json gpurun_out/r2g/bench_ws.json; tail -3 gpurun_out/r2g/pmc_sq.log
