#!/bin/bash
# round 2, GPU session J: echo chain as a parallel graph branch: parity subset, shard emulation (deterministic on/off), then the profile collection
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2j
timeout 1500 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py tests/test_hip_scene.py -m gpu -x -q -k "not alternate" > gpurun_out/r2j/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2j/summary.txt
timeout 900 python tools/emulate_shards.py --steps 20 > gpurun_out/r2j/shards_fast.log 2>&1
timeout 900 python tools/emulate_shards.py --steps 20 --deterministic > gpurun_out/r2j/shards_det.log 2>&1
bash tools/gpu_session_profile.sh r2j/prof > gpurun_out/r2j/profile.log 2>&1
cat gpurun_out/r2j/summary.txt; tail -4 gpurun_out/r2j/tests.log; grep world gpurun_out/r2j/shards_fast.log gpurun_out/r2j/shards_det.log; tail -3 gpurun_out/r2j/profile.log | cut -c1-600
