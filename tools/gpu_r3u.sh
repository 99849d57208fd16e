cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "rowgroup or groupnorm or full_eps or tiny" 2>&1 | tail -15
for v in 1 0; do
( cd /tmp && ES_GN_RG=$v timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_rg$v -o rg --output-format csv -- python $GRAFT_REPO_ROOT/tools/emulate_shards.py --worlds 1 --steps 12 2>&1 | grep "world" )
f=$(find /tmp/prof_rg$v -name '*kernel_trace.csv' | head -1)
echo "== ES_GN_RG=$v"; python tools/step_breakdown.py $f 8 | grep "step\|k_conv_ws\|k_gn\|rowgroup"
done
for v in 1 0 1 0; do echo "== ES_GN_RG=$v"; ES_GN_RG=$v timeout 600 python tools/emulate_shards.py --steps 20 --worlds 1,8 2>&1 | grep world; done
