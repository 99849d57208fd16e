cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/scene_sizes_latency.py 2>&1 | grep "O ="
