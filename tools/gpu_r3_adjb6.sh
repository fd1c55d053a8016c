#!/bin/bash
# backward: the GEMM-shaped piece on the bf16 pipe (adj_basis6_kernel) and the 16-byte target-map fold: tests, then the training step A/B
OUT=gpurun_out/r3_adjb6; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_dense_backward.py -x -q 2>&1 | tail -4 ) | tee $OUT/tests.log
for bits in 0 67108864; do
  echo "== PBITS $bits" | tee -a $OUT/train.log
  PBITS=$bits timeout 300 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep -v amdgpu | tail -4 | tee -a $OUT/train.log
  PBITS=$bits timeout 300 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -2 | tee -a $OUT/train.log
done
