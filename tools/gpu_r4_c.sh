#!/bin/bash
# round-4 session C: full GPU suite (fp32-operand route, output-conv kernel, f16 intermediates) + bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4_c}
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -3; grep -E "^FAILED|fp32-operand|fp16-operand route" $OUT/tests_gpu.log | cut -c1-220
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print(d['value'], d['value_min_max'], 'shape', d['config']['shape']['ms_per_step'], d['config']['shape']['ms_per_step_min_max'], 'layout', d['config']['layout']['ms_per_step'], 'roof', d['roofline']['achieved'], d['roofline']['avg_launch_us_min_max'], d.get('note'))
for r in d['sub_records']: print(str(r.get('config'))[:60], r.get('value'), r.get('ms_per_step'), r.get('roofline',{}).get('frac') if r.get('roofline') else None, r.get('fp16_product_vs_fp32_route_eps'), r.get('error'))
PY
tail -3 $OUT/bench.err
