#!/bin/bash
# round 5, session 1: where does a rows launch spend its time?
mkdir -p gpurun_out/s1
./tools/probes/bin/probe_l2_persist > gpurun_out/s1/probe_l2.txt 2>&1
ES_LIB_TAG=_stamp timeout 600 python tools/rows_stamps.py 32 > gpurun_out/s1/rows_stamps.txt 2>&1
for i in 1 2; do
timeout 600 python bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline --no-sub-records > gpurun_out/s1/layout_base_$i.json 2>gpurun_out/s1/layout_base_$i.err
HIP_FORCE_DEV_KERNARG=1 timeout 600 python bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline --no-sub-records > gpurun_out/s1/layout_devkarg1_$i.json 2>/dev/null
HIP_FORCE_DEV_KERNARG=0 timeout 600 python bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline --no-sub-records > gpurun_out/s1/layout_devkarg0_$i.json 2>/dev/null
done
timeout 900 python tools/layout_op_times.py 32 > gpurun_out/s1/layout_op_times.txt 2>&1
tail -3 gpurun_out/s1/probe_l2.txt; tail -5 gpurun_out/s1/rows_stamps.txt; cat gpurun_out/s1/layout_*.json | cut -c1-300
