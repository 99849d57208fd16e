#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2x
timeout 300 python tools/microbench_attention.py 2>&1 | grep -v amdgpu > gpurun_out/r2x/att.log
timeout 600 python -m pytest tests/test_hip_vol.py -m gpu -x -q -k "attention or unet3d_full_eps or vqvae" > gpurun_out/r2x/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2x/att.log
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r2x/pmc -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/microbench_attention.py > /dev/null 2>&1 )
python - <<'P' >> gpurun_out/r2x/att.log
import csv, glob, collections
d=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r2x/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attention' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': d[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in d.items(): print(k, 'FETCH_SIZE raw KB avg', sum(v)/len(v), 'n', len(v))
P
rm -rf gpurun_out/r2x/pmc
timeout 600 python bench.py --no-cpu-baseline --no-sub-records 2>/dev/null | cut -c1-260 >> gpurun_out/r2x/att.log
cat gpurun_out/r2x/att.log; tail -2 gpurun_out/r2x/tests.log
