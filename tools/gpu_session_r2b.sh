#!/bin/bash
# round 2, GPU session B: rows-kernel latency restructure + ws epilogue prefetch: parity, then timings
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
python -m pytest tests/test_hip_rows.py tests/test_hip_vol.py tests/test_hip_traj.py -m gpu -x -q > gpurun_out/r2b/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2b/summary.txt
python tools/microbench_rows.py > gpurun_out/r2b/rows.log 2>&1
python tools/microbench_power.py > gpurun_out/r2b/power.log 2>&1
python bench.py --workload layout --no-cpu-baseline > gpurun_out/r2b/bench_layout.json 2> gpurun_out/r2b/bench_layout.err
python bench.py --no-cpu-baseline > gpurun_out/r2b/bench_full.json 2> gpurun_out/r2b/bench_full.err
cat gpurun_out/r2b/summary.txt; tail -3 gpurun_out/r2b/tests.log; cat gpurun_out/r2b/rows.log gpurun_out/r2b/power.log; cat gpurun_out/r2b/bench_layout.json gpurun_out/r2b/bench_full.json | cut -c1-600
