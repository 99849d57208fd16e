#!/bin/bash
# round 6, session 13: k_conv_ws3 with ONE barrier per (chunk, kd, kh) group (ES_CONV_GB=1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s13}
mkdir -p $OUT
ES_CONV_GB=1 timeout 600 python -m pytest tests/test_hip_vol.py -q -m gpu -n 1 -k "unet3d_full_eps or shards_equal or test_conv_mfma or rowgroup" > $OUT/pytest_gb.txt 2>&1; tail -3 $OUT/pytest_gb.txt | cut -c1-200
for v in 0 1 0 1; do
ES_CONV_GB=$v timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sub-records > $OUT/bench_gb$v.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('$OUT/bench_gb$v.json') if l.startswith('{')][-1]);print('GB $v', d['value'], d['config']['shape']['ms_per_step'], d['roofline']['avg_launch_us'])"
done
