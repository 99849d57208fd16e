#!/bin/bash
# the whole GPU suite + smoke (as the driver runs them)
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests/ -q -m gpu > gpurun_out/full/gputest.txt 2>&1; tail -15 gpurun_out/full/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.txt 2>&1; tail -3 gpurun_out/full/smoke.txt
