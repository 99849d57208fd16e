#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2ac
timeout 1200 python -m pytest tests/test_hip_vol.py -m gpu -x -q > gpurun_out/r2ac/tests.log 2>&1
echo "tests rc=$?" > gpurun_out/r2ac/summary.txt
for v in 0 1 0 1; do
  echo "WSSPLIT=$v" >> gpurun_out/r2ac/summary.txt
  ES_CONV_WSSPLIT=$v timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" >> gpurun_out/r2ac/summary.txt
  ES_CONV_WSSPLIT=$v timeout 600 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['config']['shape']['ms_per_step'])" >> gpurun_out/r2ac/summary.txt
done
cat gpurun_out/r2ac/summary.txt; tail -3 gpurun_out/r2ac/tests.log
