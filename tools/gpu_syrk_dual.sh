#!/bin/bash
# NOTE: the reserved_ bit this script toggles belongs to an experiment that is no longer in the library (its code: see
# banet_amd/csrc/experiments/README.md and the *.patch.txt / *.hip.txt files there); kept as the record of how the numbers were taken.
# A/B of ba_syrk_dual_kernel (reserved_ bit 20) against the one-role SYRK: kernel times + bit-identity (prof_assemble.py), then the bench
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
PB=32 PBITS=0,1048576 PROUNDS=3 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee $OUT/sd_a.log
PB=8 PBITS=0,1048576 PROUNDS=3 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee $OUT/sd_b.log
PB=32 PN=50 PBITS=0,1048576 PROUNDS=2 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee $OUT/sd_c.log
for bits in 0 1048576 0 1048576; do
  timeout 600 python bench.py --steps 4 --warmup 2 --no-sweep --no-cpu-baseline --reserved $bits 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('reserved', $bits, 'value', j['value'], 'ms', j['ms_per_step'], 'parity', j.get('parity'))
" | cut -c1-300 | tee -a $OUT/sd_bench.log
done
exit 0
