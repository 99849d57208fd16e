cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_hip_rows.py -x -q > gpurun_out/r3a/rows_tests.log 2>&1
echo "rows tests rc=$?"
tail -5 gpurun_out/r3a/rows_tests.log
timeout 600 python tools/microbench_rows.py > gpurun_out/r3a/rows_microbench.txt 2>&1
grep -v amdgpu gpurun_out/r3a/rows_microbench.txt
for SP in 256 128; do for LN in 1 2; do
ES_ROWS_SPLIT=$SP ES_ROWS_LN_SPLIT=$LN timeout 600 python bench.py --workload layout --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r3a/bench_layout_${SP}_${LN}.json 2> gpurun_out/r3a/bench_layout.err
echo "split=$SP ln=$LN: $(tail -1 gpurun_out/r3a/bench_layout_${SP}_${LN}.json | cut -c1-160)"
done; done
ES_ROWS_SPLIT=0 timeout 600 python bench.py --workload layout --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
