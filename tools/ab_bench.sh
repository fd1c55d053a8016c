#!/bin/bash
# same-box bench-level A/B: in-tree build (reserved 0 / 65536 = packed patch) vs banet_amd/lib_ab/libbanet_hip_old.so
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
OLD=$PWD/banet_amd/lib_ab/libbanet_hip_old.so
for rep in 1 2; do
  for cfg in "new 0" "new 65536" "old 0"; do
    set -- $cfg
    if [ $1 = old ]; then export BANET_HIP_LIB=$OLD; else unset BANET_HIP_LIB; fi
    timeout 600 python bench.py --steps 4 --warmup 2 --no-sweep --no-parity --no-cpu-baseline --reserved $2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$1', 'reserved', $2, 'value', j['value'], 'ms', j['ms_per_step'], {k: v['gather_avg_us'] for k, v in j['roofline']['per_level'].items()})
"
  done
done 2>&1 | tee $OUT/ab_bench.log
exit 0
