#!/bin/bash
# round 6, session 7: what the fp32x step is made of; 1x1 launches of the 32-object step on 64- / 128-row tiles (forced)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s7}
mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_fp32x -o x --output-format csv -- python $GRAFT_REPO_ROOT/tools/profile_fp32x.py 32 fp32x > $OUT/prof_fp32x.log 2>&1 )
KS=$(find $OUT/prof_fp32x -name "*kernel_stats.csv" | head -1); head -25 $KS | cut -c1-200
tail -2 $OUT/prof_fp32x.log
ES_TOOL_VOL_OPTIONS=conv_st_bm=64 timeout 300 python tools/conv_launch_table.py 32 2>&1 | grep -v amdgpu > $OUT/conv_table_O32_bm64.txt
ES_TOOL_VOL_OPTIONS=conv_st_bm=128,conv_st_np=8,conv_st_ns=5 timeout 300 python tools/conv_launch_table.py 32 2>&1 | grep -v amdgpu > $OUT/conv_table_O32_bm128.txt
timeout 300 python tools/conv_launch_table.py 32 2>&1 | grep -v amdgpu > $OUT/conv_table_O32.txt
timeout 300 python tools/conv_launch_table.py 16 2>&1 | grep -v amdgpu > $OUT/conv_table_O16.txt
ES_TOOL_VOL_OPTIONS=conv_few=0 timeout 300 python tools/conv_launch_table.py 16 2>&1 | grep -v amdgpu > $OUT/conv_table_O16_few0.txt
find $OUT -name "*.db" -delete; find $OUT -name "*.rocpd" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +6M -delete
head -2 $OUT/conv_table_O*.txt
