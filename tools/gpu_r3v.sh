cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 0; do
( cd /tmp && ES_GN_RG=$v timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_rg$v -o rg --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --worlds 1 --steps 12 2>&1 | grep "world" )
f=$(find /tmp/prof_rg$v -name '*kernel_trace.csv' | head -1)
echo "== ES_GN_RG=$v"; python tools/step_breakdown.py $f 8 | head -24
done
