#!/bin/bash
# device-side weight re-layouts: identity tests, the first call of a process, then the whole GPU suite
tag=${1:-pack}
out=gpurun_out/$tag
mkdir -p $out
timeout 600 python -m pytest tests/test_hip_rows.py tests/test_hip_vol.py -m gpu -x -q -k "relayout" > $out/tests_pack.log 2>&1
echo "pack tests rc=$?" > $out/summary.txt
timeout 900 python tools/e2e_latency.py --profile-first > $out/e2e_latency.txt 2>&1
echo "e2e rc=$?" >> $out/summary.txt
timeout 600 python tools/scene_sizes_latency.py > $out/scene_sizes.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1
echo "gpu tests rc=$?" >> $out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1
echo "smoke rc=$?" >> $out/summary.txt
tail -3 $out/tests_pack.log; grep -v "^ *[0-9]* *[0-9.]* *[0-9.]* *[0-9.]* *[0-9.]* {" $out/e2e_latency.txt | tail -40; cat $out/scene_sizes.txt; tail -3 $out/tests_gpu.log; cat $out/summary.txt
