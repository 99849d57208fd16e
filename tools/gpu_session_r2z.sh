#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 60 tools/probes/probe_ds_read_tr > gpurun_out/r2z/tr.log 2>&1
head -40 gpurun_out/r2z/tr.log
