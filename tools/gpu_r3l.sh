cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "0 2" "0 1" "0 3"; do set -- $cfg
for i in 1 2; do
ES_ROWS_SKIP_EARLY=$1 ES_ROWS_GCN_SLICES=$2 timeout 600 python bench.py --workload layout --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-150 | sed "s/^/skip_early=$1 gcn_slices=$2 /"
done; done
timeout 900 python tools/layout_op_times.py 2>&1 | grep -v amdgpu | head -16
