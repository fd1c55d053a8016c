#!/bin/bash
# round 6, first GPU call: the target-tile backward (tests) + the training step A/B (fold 8x8 / fold 8x4 / round-5 path), same box
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_dense_backward.py -m gpu -q --timeout 900 -p no:cacheprovider -rfE --tb=short -x ) 2>&1 | tail -25 | tee $OUT/r6a_tests.txt
for mode in 1 2 0 1 0; do
  echo "BANET_ADJOINT_FOLD=$mode"
  BANET_ADJOINT_FOLD=$mode timeout 600 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep -v amdgpu | tail -3
done | tee $OUT/r6a_dense_train.txt
for mode in 1 0; do
  echo "BANET_ADJOINT_FOLD=$mode 8 windows"
  BANET_ADJOINT_FOLD=$mode timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3
  echo "BANET_ADJOINT_FOLD=$mode 8 five-frame windows"
  PFRAMES=5 BANET_ADJOINT_FOLD=$mode timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3
done | tee -a $OUT/r6a_dense_train.txt
exit 0
