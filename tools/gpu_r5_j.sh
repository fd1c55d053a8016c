#!/bin/bash
# Round 5, extras on the final build: the dense training step (forward + fused backward) at 32 and 8 windows, the fused sparse
# training iteration at the reference's shape, the sparse inference path (tools/bench_sparse.py).
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for w in 32 8; do timeout 600 python tools/bench_dense_train.py $w 480 640 2 2>&1 | grep -v amdgpu | tail -4; done | tee $OUT/r5j_dense_train.txt
PB=4 PN=4096 PH=384 PW=512 timeout 600 python tools/train_graph_bench.py 2>&1 | grep -v amdgpu | tee $OUT/r5j_train_graph.txt
timeout 600 python tools/bench_sparse.py 2>&1 | grep -v amdgpu | tail -8 | tee $OUT/r5j_bench_sparse.txt
exit 0
