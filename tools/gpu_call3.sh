#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_dense_backward.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) > $OUT/pytest_bwd.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_bwd.log
grep -v "^$" $OUT/pytest_bwd.log | tail -60
exit 0
