cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof8 -o w8 --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --worlds 8 --steps 20 2>&1 | grep "^world" )
f=$(find /tmp/prof8 -name '*kernel_trace.csv' | head -1)
python tools/step_breakdown.py $f 10 > gpurun_out/w8_step_breakdown.txt; head -16 gpurun_out/w8_step_breakdown.txt
timeout 600 python tools/conv_launch_table.py 4 2>&1 | grep -v amdgpu > gpurun_out/conv_table_O4.txt; head -3 gpurun_out/conv_table_O4.txt
