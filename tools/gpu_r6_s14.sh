#!/bin/bash
# round 6, session 14: k_conv_ws3 B-fragment prefetch three columns ahead (product) against two (libechoscene_hip_pf2.so)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s14}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_vol.py -q -m gpu -n 1 -k "unet3d_full_eps or shards_equal or test_conv_mfma or rowgroup" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt | cut -c1-200
for v in _pf2 "" _pf2 ""; do
ES_LIB_TAG=$v timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-sub-records > $OUT/bench$v.json 2>/dev/null; python -c "
import json;d=json.loads([l for l in open('$OUT/bench$v.json') if l.startswith('{')][-1]);print('lib [$v]', d['value'], d['config']['shape']['ms_per_step'], d['roofline']['avg_launch_us'])"
done
