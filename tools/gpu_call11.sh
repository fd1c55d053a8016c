#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_dense_backward.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -k "sample_stats or training or backward or adjoint or differentiable or resize" ) > $OUT/pytest_det.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_det.log
grep -v "^$" $OUT/pytest_det.log | tail -20
PB=4 PH=120 PW=160 timeout 300 python tools/train_graph_bench.py 2>&1 | tail -3
exit 0
