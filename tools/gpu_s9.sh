#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s11
mkdir -p $OUT
# page the image in first (torch, its code objects, our library): every measurement below is a NEW process on a warm file cache
timeout 300 python tools/e2e_latency.py --nodes 8 > $OUT/warmup.txt 2>&1; grep -E "^call" $OUT/warmup.txt
for v in plain host prewarm plain2 host2; do
  case $v in plain|plain2) F="";; host|host2) F="--host-weights";; prewarm) F="--prewarm 12";; esac
  timeout 200 python tools/e2e_latency.py $F > $OUT/e2e_$v.txt 2>&1
  echo "== $v"; grep -E "^call|prewarm" $OUT/e2e_$v.txt
done
