cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 256 128 64 0; do echo "== ES_ROWS_SPLIT=$v"; ES_ROWS_SPLIT=$v timeout 900 python tools/e2e_latency.py 2>&1 | grep "call\|loop\|decode"; done
