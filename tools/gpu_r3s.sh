cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep world
timeout 1200 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "conv_mfma or full_eps or shards_equal or alternate" 2>&1 | tail -3
