#!/bin/bash
mkdir -p gpurun_out/s4
B="python bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline --no-sub-records"
for i in 1 2; do
timeout 600 $B > gpurun_out/s4/base_$i.json 2>gpurun_out/s4/base_$i.err
ES_X_NT=2 timeout 600 $B > gpurun_out/s4/nt2_$i.json 2>/dev/null
ES_X_XCD=1 timeout 600 $B > gpurun_out/s4/xcd_$i.json 2>/dev/null
ES_X_XCD=1 ES_X_NT=2 timeout 600 $B > gpurun_out/s4/xcd_nt2_$i.json 2>/dev/null
done
ES_LIB_TAG=_stamp timeout 600 python tools/rows_stamps.py 32 > gpurun_out/s4/rows_stamps.txt 2>&1
ES_X_XCD=1 ES_LIB_TAG=_stamp timeout 600 python tools/rows_stamps.py 32 > gpurun_out/s4/rows_stamps_xcd.txt 2>&1
timeout 900 python -m pytest tests/test_hip_rows.py -q > gpurun_out/s4/test_rows.txt 2>&1; tail -4 gpurun_out/s4/test_rows.txt
ES_X_XCD=1 ES_X_NT=2 timeout 900 python -m pytest tests/test_hip_rows.py -q > gpurun_out/s4/test_rows_sw.txt 2>&1; tail -4 gpurun_out/s4/test_rows_sw.txt
for f in gpurun_out/s4/*.json; do echo $f $(cut -c95-180 $f); done
tail -4 gpurun_out/s4/rows_stamps.txt | cut -c1-300
