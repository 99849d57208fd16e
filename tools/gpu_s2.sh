#!/bin/bash
# round 5, session 2: k_rows_frag -- correctness, timeline, A/B against k_linear_rows
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests/test_hip_rows.py -x -q > gpurun_out/s2/test_rows.txt 2>&1; tail -5 gpurun_out/s2/test_rows.txt
ES_LIB_TAG=_stamp timeout 600 python tools/rows_stamps.py 32 > gpurun_out/s2/rows_stamps.txt 2>&1
for i in 1 2; do
for f in 0 1; do
timeout 600 python tools/with_rows_family.py $f bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline --no-sub-records > gpurun_out/s2/layout_family${f}_$i.json 2>gpurun_out/s2/layout_family${f}_$i.err
done; done
timeout 900 python -m pytest tests/test_hip_traj.py -x -q -k layout > gpurun_out/s2/test_traj.txt 2>&1; tail -3 gpurun_out/s2/test_traj.txt
tail -8 gpurun_out/s2/rows_stamps.txt | cut -c1-400; cat gpurun_out/s2/layout_*.json | cut -c1-200
