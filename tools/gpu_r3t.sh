cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/aux_launch_table.py 2>&1 | tail -40
