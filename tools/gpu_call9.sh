#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_dense_backward.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) > $OUT/pytest_bwd.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_bwd.log
grep -v "^$" $OUT/pytest_bwd.log | tail -20
timeout 600 python tools/bench_dense_train.py 8 480 640 2 > $OUT/dense_train.log 2>&1; tail -5 $OUT/dense_train.log
BANET_TRAIN_GRAPH=0 timeout 600 python tools/bench_dense_train.py 8 480 640 2 > $OUT/dense_train_eager.log 2>&1; tail -5 $OUT/dense_train_eager.log
timeout 600 python tools/bench_dense_train.py 32 480 640 2 > $OUT/dense_train_b32.log 2>&1; tail -5 $OUT/dense_train_b32.log
exit 0
