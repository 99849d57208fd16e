cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $OUT
timeout 600 python tools/probe_step_graph.py > $OUT/probe32.txt 2>&1; grep -i "captured\|sharded-structure\|STEP_GRAPH\|Error\|warn" $OUT/probe32.txt | head
timeout 900 python tools/probe_step_graph.py 224 > $OUT/probe224.txt 2>&1; grep -i "captured\|sharded-structure\|STEP_GRAPH\|Error\|warn" $OUT/probe224.txt | head
timeout 1200 python -m pytest tests/test_hip_scene.py -x -q -k "model_files or captured_graph" > $OUT/tests.log 2>&1; tail -25 $OUT/tests.log
