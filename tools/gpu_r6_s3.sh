#!/bin/bash
# round 6, session 3: the few-objects routes in the dispatcher + the canonical (deterministic) arithmetic
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s3}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_vol.py -x -q -m gpu -k "shards or canonical or few_objects or unet3d_full_eps or test_conv_mfma or rowgroup or ddim" > $OUT/pytest_vol.txt 2>&1
tail -12 $OUT/pytest_vol.txt
timeout 600 python tools/emulate_shards.py --steps 20 > $OUT/emu_tuned.txt 2>&1
timeout 600 python tools/emulate_shards.py --steps 20 --deterministic > $OUT/emu_exact.txt 2>&1
ES_TOOL_VOL_OPTIONS=conv_few=0 timeout 600 python tools/emulate_shards.py --steps 20 --worlds 1,8 > $OUT/emu_tuned_few0.txt 2>&1
grep world $OUT/emu_tuned.txt $OUT/emu_exact.txt $OUT/emu_tuned_few0.txt
timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32.txt 2>&1
ES_TOOL_VOL_OPTIONS=conv_few=0 timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32_few0.txt 2>&1
ES_CONV_NS=4 timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32_ns4.txt 2>&1
ES_CONV_NS=5 timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32_ns5.txt 2>&1
ES_LIN_NCB_MAX=2 timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32_ncb2.txt 2>&1
ES_LIN_NCB_MAX=4 timeout 300 python tools/conv_launch_table.py 32 > $OUT/conv_table_O32_ncb4.txt 2>&1
head -3 $OUT/conv_table_O32*.txt
timeout 300 python tools/shard_op_table.py --world 8 > $OUT/op_table_w8_tuned.txt 2>&1
head -14 $OUT/op_table_w8_tuned.txt
