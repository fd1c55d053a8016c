#!/bin/bash
# backward tests + training-step timings (tools/bench_dense_train.py) at 8 and 32 windows
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_dense_backward.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short ) 2>&1 | tail -4
for w in 2 8 32; do timeout 600 python tools/bench_dense_train.py $w 480 640 2 2>&1 | grep -v amdgpu | tail -4; done | tee $OUT/dense_train.log
BANET_SPD_SOLVE=0 timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3 | tee $OUT/dense_train_torch_solve.log
exit 0
