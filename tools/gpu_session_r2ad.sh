#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2ad
timeout 1500 python -m pytest tests/test_hip_vol.py -m gpu -x -q > gpurun_out/r2ad/tests.log 2>&1
echo "vol tests rc=$?" > gpurun_out/r2ad/summary.txt
timeout 1500 python -m pytest tests/test_hip_scene.py tests/test_hip_traj.py -m gpu -x -q > gpurun_out/r2ad/tests2.log 2>&1
echo "scene/traj tests rc=$?" >> gpurun_out/r2ad/summary.txt
timeout 600 python tools/emulate_shards.py --steps 20 --deterministic 2>&1 | grep "^world" >> gpurun_out/r2ad/summary.txt
cat gpurun_out/r2ad/summary.txt; tail -3 gpurun_out/r2ad/tests.log; tail -5 gpurun_out/r2ad/tests2.log
