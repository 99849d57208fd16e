cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
for D in 4 11 3; do ES_ROWS_DBG=$D timeout 600 python tools/microbench_rows.py cold 2>&1 | grep "^cold" | grep "plain512\|gn_silu1024"; done
for SP in 128 512; do
ES_ROWS_SPLIT=$SP timeout 600 python bench.py --workload layout --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200
done
