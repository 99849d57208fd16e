#!/bin/bash
mkdir -p gpurun_out/s3
ES_DEBUG_SYNC=1 timeout 300 python -m pytest tests/test_hip_rows.py -x -q -k "gcn_vs_reference_golden and res_bn" > gpurun_out/s3/dbg_gcn.txt 2>&1; grep "^\[es\]\|passed\|failed\|fault\|err" gpurun_out/s3/dbg_gcn.txt | tail -12
timeout 300 python -m pytest tests/test_hip_rows.py -x -q -k "multi_problem or folded_rider" > gpurun_out/s3/dbg_multi.txt 2>&1; tail -15 gpurun_out/s3/dbg_multi.txt | cut -c1-200
