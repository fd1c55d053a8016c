#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=8 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -22 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 ) > $OUT/bench.log 2>&1; echo "bench exit $?" >> $OUT/bench.log
tail -c 600 $OUT/bench.log
exit 0
