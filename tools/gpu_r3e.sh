cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
timeout 900 python tools/conv_launch_table.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3e/conv_launch_table.txt
timeout 300 python tools/step_breakdown.py 2>&1 | grep -v amdgpu | tail -30 | tee gpurun_out/r3e/step_breakdown.txt
