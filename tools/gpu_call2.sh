#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/test_gpu_round2.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -s ) > $OUT/pytest_gpu2.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu2.log
grep -v "^$" $OUT/pytest_gpu2.log | tail -30
( time timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 ) > $OUT/bench.log 2>&1; echo "bench exit $?" >> $OUT/bench.log
tail -c 1500 $OUT/bench.log
exit 0
