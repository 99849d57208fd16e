#!/bin/bash
mkdir -p gpurun_out/s6
B="python bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline --no-sub-records"
timeout 900 python -m pytest tests/test_hip_rows.py -x -q > gpurun_out/s6/test_rows.txt 2>&1; tail -4 gpurun_out/s6/test_rows.txt
for i in 1 2; do
timeout 600 $B > gpurun_out/s6/pre_$i.json 2>gpurun_out/s6/pre_$i.err
ES_ROWS_PREFETCH=0 timeout 600 $B > gpurun_out/s6/nopre_$i.json 2>gpurun_out/s6/nopre_$i.err
done
ES_LIB_TAG=_stamp timeout 600 python tools/rows_stamps.py 32 > gpurun_out/s6/rows_stamps.txt 2>&1
timeout 900 python -m pytest tests/test_hip_traj.py -x -q -k layout > gpurun_out/s6/test_traj.txt 2>&1; tail -3 gpurun_out/s6/test_traj.txt
for f in gpurun_out/s6/*.json; do echo $f $(cut -c95-200 $f); done
tail -5 gpurun_out/s6/rows_stamps.txt | cut -c1-300
