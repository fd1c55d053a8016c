#!/bin/bash
# backward without zero-fills (BANET_ADJOINT_OVERWRITE on the first call of a level): tests, then the training step
OUT=gpurun_out/r3_ow; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_dense_backward.py -x -q 2>&1 | tail -4 ) | tee $OUT/tests.log
timeout 300 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep -v amdgpu | tail -4 | tee -a $OUT/train.log
timeout 300 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -2 | tee -a $OUT/train.log
PFRAMES=5 timeout 300 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3 | tee -a $OUT/train.log
