cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_rows.py -x -q > $OUT/rows_tests.log 2>&1
echo "rows tests rc=$?"; tail -4 $OUT/rows_tests.log
for D in 0 2; do ES_ROWS_DBG=$D timeout 600 python tools/microbench_rows.py cold 2>&1 | grep "^cold"; done
for LN in 1 2; do
ES_ROWS_LN_SPLIT=$LN timeout 600 python bench.py --workload layout --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200
done
