cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3h
echo "=== baseline"; timeout 600 python tools/conv_launch_table.py 2>&1 | grep -v amdgpu | grep "taps  1\|summed" 
echo "=== ES_CONV_PP=-1 (two 128-row WGs per CU, no stagger)"; ES_CONV_PP=-1 timeout 600 python tools/conv_launch_table.py 2>&1 | grep -v amdgpu | grep "taps  1\|summed"
echo "=== ES_CONV_PP=0 (auto stagger)"; ES_CONV_PP=0 timeout 600 python tools/conv_launch_table.py 2>&1 | grep -v amdgpu | grep "taps  1\|summed"
echo "=== ES_CONV_PP=12 (12 us)"; ES_CONV_PP=12 timeout 600 python tools/conv_launch_table.py 2>&1 | grep -v amdgpu | grep "taps  1\|summed"
echo "=== ES_CONV_PP3=0 (3x3x3 too, auto stagger)"; ES_CONV_PP3=0 timeout 600 python tools/conv_launch_table.py 2>&1 | grep -v amdgpu | grep "summed\|@16x16x16" | head -12
