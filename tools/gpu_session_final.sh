#!/bin/bash
# full GPU test suite, smoke, then the profile collection of the round
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-final}
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q -s > $OUT/tests_gpu.log 2>&1
echo "gpu tests rc=$?" > $OUT/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/summary.txt
timeout 300 python tools/e2e_latency.py > $OUT/e2e.txt 2>&1
timeout 400 python tools/scene_sizes_latency.py > $OUT/scene_sizes.txt 2>&1
timeout 300 python tools/aux_launch_table.py > $OUT/aux_table.txt 2>&1
bash tools/gpu_session_profile.sh ${1:-final}
timeout 600 python tools/emulate_shards.py --steps 20 2>&1 | grep "^world" > $OUT/shards_default.txt
timeout 600 python tools/emulate_shards.py --steps 20 --tuned 2>&1 | grep "^world" > $OUT/shards_tuned.txt

timeout 300 python tools/model_file_size.py > $OUT/model_file_size.txt 2>&1
cat $OUT/summary.txt; grep -E "passed|failed" $OUT/tests_gpu.log | tail -3; tail -2 $OUT/smoke.log; grep -v amdgpu $OUT/e2e.txt | tail -8; grep -v amdgpu $OUT/scene_sizes.txt; cat $OUT/shards_default.txt $OUT/shards_tuned.txt; grep -v amdgpu $OUT/model_file_size.txt; du -sh $OUT
