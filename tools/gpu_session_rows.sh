#!/bin/bash
# rows-path GPU tests (GCN poolings, model files replayed by the C host) + first-call breakdown
tag=${1:-rows}
out=gpurun_out/$tag
mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_rows.py tests/test_hip_scene.py -m gpu -x -q > $out/tests_rows.log 2>&1
echo "rows/scene tests rc=$?" > $out/summary.txt
tail -3 $out/tests_rows.log
cat $out/summary.txt
