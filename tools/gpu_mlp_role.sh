#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
for W in 1 8; do
  for R in 0 32768 0 32768; do
    timeout 600 python bench.py --windows $W --steps 4 --warmup 2 --no-sweep --no-parity --no-cpu-baseline --reserved $R 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('windows', $W, 'reserved', $R, 'value', j['value'], 'ms_per_step', j['ms_per_step'], {k: (v['syrk_avg_us'], v.get('level_ms_last_step')) for k, v in j['roofline']['per_level'].items()})
"
  done
done 2>&1 | tee $OUT/mlp_role_ab.log
exit 0
