cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/linear_stamps.py 2>&1 | tail -30
