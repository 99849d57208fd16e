#!/bin/bash
# round 6, session 10: k_conv_ws3 on the few-objects 128-row tile; default-on for the 256-row tile
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6_s10}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_vol.py -q -m gpu -n 1 -k "few_objects or shards or canonical or unet3d_full_eps or test_conv_mfma or rowgroup" > $OUT/pytest_sel.txt 2>&1; tail -5 $OUT/pytest_sel.txt | cut -c1-250
ES_CONV_A3=0 timeout 400 python tools/emulate_shards.py --steps 20 --worlds 1,4,8 2>&1 | grep "^world" | sed 's/^/A3 off: /'
timeout 400 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" | sed 's/^/A3 on:  /'
timeout 400 python tools/emulate_shards.py --steps 20 --deterministic --worlds 1,8 2>&1 | grep "^world" | sed 's/^/A3 on, canonical: /'
