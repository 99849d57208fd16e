cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_vol.py -x -q > $OUT/vol.log 2>&1; echo "vol rc=$?"; tail -3 $OUT/vol.log
timeout 600 python tools/conv_launch_table.py 2>&1 | grep "summed\|N    3" 
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-sub-records 2>/dev/null | tail -1 | cut -c170-330
