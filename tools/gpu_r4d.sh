cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3fin4
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 $OUT/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --workload layout --steps 300 --warmup 10 > $OUT/bench_layout.json 2> $OUT/bench_layout.err; tail -1 $OUT/bench_layout.json | cut -c1-260
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_layout -o lay --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload layout --steps 50 --warmup 3 --no-cpu-baseline > $OUT/prof_layout.log 2>&1 )
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_final_nocpu.json; python -c "
import json; d=json.loads(open('$OUT/bench_final_nocpu.json').read()); print('full %.3f ms  value %.3f  shape %.3f  layout %.4f  frac %.4f' % (d['ms_per_step'], d['value'], d['config']['shape']['ms_per_step'], d['config']['layout']['ms_per_step'], d['roofline']['frac']))"
timeout 300 python tools/e2e_latency.py 2>&1 | grep "call 2\|loop\|decode"
