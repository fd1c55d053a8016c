#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -k "equation or eq_ or literal or training" ) > $OUT/pytest_eq.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_eq.log
grep -v "^$" $OUT/pytest_eq.log | tail -30
timeout 600 python tools/bench_eqcon.py 2>&1 | grep -v amdgpu | tee $OUT/bench_eqcon.log
timeout 600 python tools/fuzz_parity.py 2>&1 | tail -5
exit 0
