cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_hip_vol.py tests/test_hip_traj.py "tests/test_hip_scene.py::test_bench_two_ranks_on_one_gpu_strong_and_weak" -x -q -s > $OUT/tests.log 2>&1
echo "vol+traj tests rc=$?"; grep -E "passed|failed|rel err|max abs err|IoU|Error" $OUT/tests.log | tail -30
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-200; python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r3d/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'avg us', d['roofline']['avg_launch_us'])
print('layout', d['config']['layout']['ms_per_step'], 'shape', d['config']['shape']['ms_per_step'])
for r in d.get('sub_records', []):
    print(r if not isinstance(r, dict) else {k: r[k] for k in list(r)[:6]})
PY
cp echoscene_amd/libechoscene_hip.so /tmp/lib_backup.so
ES_BUILD_FLAGS=-DES_STAMP python -m echoscene_amd.build --force > /tmp/build.log 2>&1 || tail -20 /tmp/build.log
timeout 600 python tools/conv_stamps.py 2>&1 | grep -v amdgpu | grep "==\|round 1 cons"
cp /tmp/lib_backup.so echoscene_amd/libechoscene_hip.so
