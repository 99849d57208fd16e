#!/bin/bash
# round 2, GPU session E: whole GPU suite on the cleaned kernels + plan cache + configs[4] rank shapes; bench modes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2e
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/all.log 2>&1
echo "all rc=$?" >> gpurun_out/r2e/summary.txt
python bench.py > gpurun_out/r2e/bench_full.json 2> gpurun_out/r2e/bench_full.err
python bench.py --scaling weak --no-cpu-baseline --steps 20 > gpurun_out/r2e/bench_weak.json 2> gpurun_out/r2e/bench_weak.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2e/smoke.log 2>&1
cat gpurun_out/r2e/summary.txt; tail -12 gpurun_out/r2e/all.log; cat gpurun_out/r2e/bench_full.json | cut -c1-3000; cat gpurun_out/r2e/bench_weak.json | cut -c1-900; tail -3 gpurun_out/r2e/bench_weak.err; tail -2 gpurun_out/r2e/smoke.log
