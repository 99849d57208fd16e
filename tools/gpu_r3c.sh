cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -x -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?"
grep -E "passed|failed|rel err|max abs err|IoU" $OUT/tests_gpu.log | tail -40
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 300 python tools/e2e_latency.py 2>&1 | grep -v amdgpu | tail -6
