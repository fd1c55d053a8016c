#!/bin/bash
# literal op: parity tests, then A/B of the J loads of eq_syrk_kernel (LDS image vs BANET_EQ_DIRECT_LOADS=1) on one box
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -rfE --tb=short -x -k "equation_construction" ) 2>&1 | tail -8 | tee $OUT/eq_ab.log
for rep in 1 2; do
  echo "-- LDS image" | tee -a $OUT/eq_ab.log
  EQ_SHAPES=${EQ_SHAPES:-8x76800x262,2x76800x262,8x76800x134,8x76800x200} timeout 200 python tools/bench_eqcon.py 2>&1 | grep "^B=" | tee -a $OUT/eq_ab.log
  echo "-- direct dword loads" | tee -a $OUT/eq_ab.log
  BANET_EQ_DIRECT_LOADS=1 EQ_SHAPES=${EQ_SHAPES:-8x76800x262,2x76800x262,8x76800x134,8x76800x200} timeout 200 python tools/bench_eqcon.py 2>&1 | grep "^B=" | tee -a $OUT/eq_ab.log
done
exit 0
