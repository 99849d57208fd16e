#!/bin/bash
# round 2, GPU session A: new parity tests + clock experiment on the dominant conv kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
python -m pytest tests/test_hip_traj.py -m gpu -x -q -s > gpurun_out/r2a/traj.log 2>&1
echo "traj rc=$?" >> gpurun_out/r2a/summary.txt
python -m pytest tests/test_hip_scene.py -m gpu -x -q -s > gpurun_out/r2a/scene.log 2>&1
echo "scene rc=$?" >> gpurun_out/r2a/summary.txt
python -m pytest tests -m gpu -x -q --deselect tests/test_hip_traj.py --deselect tests/test_hip_scene.py > gpurun_out/r2a/rest.log 2>&1
echo "rest rc=$?" >> gpurun_out/r2a/summary.txt
# clock: GRBM_GUI_ACTIVE / wall for k_conv_ws (same launch 20x, random vs zero operands)
python tools/microbench_power.py > gpurun_out/r2a/power_plain.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_clock -o clk --output-format csv -- python $GRAFT_REPO_ROOT/tools/microbench_power.py > $GRAFT_REPO_ROOT/gpurun_out/r2a/power_pmc.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_clock2 -o clk2 --output-format csv -- python $GRAFT_REPO_ROOT/tools/microbench_power.py > $GRAFT_REPO_ROOT/gpurun_out/r2a/power_pmc2.log 2>&1 )
ls -R gpurun_out/r2a | head -50
cat gpurun_out/r2a/summary.txt
