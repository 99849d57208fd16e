cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do
ES_ROWS_FUSE=0 timeout 600 python bench.py --workload layout --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-150 | sed 's/^/nofuse /'
timeout 600 python bench.py --workload layout --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-150 | sed 's/^/fuse   /'
done
