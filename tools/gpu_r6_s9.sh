#!/bin/bash
# round 6, session 9: k_conv_ws3 (shared A tile of a (chunk, kd, kh) group, register shifts) -- bits and time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s9}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_vol.py -q -m gpu -n 1 -k "alternate_kernels and (env9 or env10)" > $OUT/pytest_a3.txt 2>&1; tail -6 $OUT/pytest_a3.txt | cut -c1-250
ES_CONV_A3=1 timeout 600 python -m pytest tests/test_hip_vol.py -q -m gpu -n 1 -k "unet3d_full_eps or shards_equal or test_conv_mfma" > $OUT/pytest_a3b.txt 2>&1; tail -4 $OUT/pytest_a3b.txt | cut -c1-250
ES_CONV_A3=0 timeout 300 python tools/conv_launch_table.py 32 2>&1 | grep -v amdgpu > $OUT/conv_table_O32_a3off.txt
ES_CONV_A3=1 timeout 300 python tools/conv_launch_table.py 32 2>&1 | grep -v amdgpu > $OUT/conv_table_O32_a3on.txt
head -1 $OUT/conv_table_O32_a3off.txt $OUT/conv_table_O32_a3on.txt
ES_CONV_A3=0 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sub-records > $OUT/bench_a3off.json 2>/dev/null; tail -1 $OUT/bench_a3off.json | cut -c1-20; python -c "
import json;d=json.loads([l for l in open('$OUT/bench_a3off.json') if l.startswith('{')][-1]);print('A3 off', d['value'], d['config']['shape']['ms_per_step'], d['roofline']['avg_launch_us'])"
ES_CONV_A3=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sub-records > $OUT/bench_a3on.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('$OUT/bench_a3on.json') if l.startswith('{')][-1]);print('A3 on', d['value'], d['config']['shape']['ms_per_step'], d['roofline']['avg_launch_us'])"
