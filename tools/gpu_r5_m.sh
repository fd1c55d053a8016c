#!/bin/bash
# kernel list of ONE fused sparse training iteration (forward + backward) under rocprofv3 --kernel-trace --stats
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
rm -rf /tmp/prof_sp && mkdir -p /tmp/prof_sp
(cd /tmp && BANET_TRAIN_GRAPH=0 PGRAPHS=fused PB=4 PN=4096 PH=384 PW=512 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sp -o sp -- python $REPO/tools/train_graph_bench.py > $REPO/$OUT/r5m_run.log 2>&1)
for f in $(find /tmp/prof_sp -name "*kernel_stats.csv"); do cp "$f" $OUT/r5m_sparse_train_kernel_stats.csv; done
grep -v amdgpu $OUT/r5m_run.log | tail -n 3
head -n 40 $OUT/r5m_sparse_train_kernel_stats.csv | cut -c1-150
exit 0
