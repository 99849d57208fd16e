#!/bin/bash
# same-box A/B: previous revision's library (ES_LIB_TAG=_prev) against the tree's, layout line twice each; then the GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s12
mkdir -p $OUT
for i in 1 2; do
  ES_LIB_TAG=_prev python bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline > $OUT/lay_prev_$i.json 2> $OUT/lay_prev_$i.err
  python bench.py --workload layout --steps 1000 --warmup 20 --no-cpu-baseline > $OUT/lay_new_$i.json 2> $OUT/lay_new_$i.err
done
grep -h -o '"ms_per_step": [0-9.]*' $OUT/lay_prev_*.json $OUT/lay_new_*.json
timeout 2400 python -m pytest tests/ -q -m gpu > $OUT/gputest.txt 2>&1; tail -4 $OUT/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
