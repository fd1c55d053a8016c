#!/bin/bash
# same-box A/B (in-tree build vs banet_amd/lib_ab/libbanet_hip_old.so): parity tests on the new build, then prof_assemble at
# 640x480 x 32 / x 8, 320x240 x 32 and 5-frame windows, alternating the two libraries
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
OLD=$PWD/banet_amd/lib_ab/libbanet_hip_old.so
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) 2>&1 | tail -5 | tee $OUT/abq_tests.log
for rep in 1 2; do
  for cfg in "PB=32" "PB=8" "PB=32 PH=240 PW=320" "PB=8 PP=4"; do
    for lib in new old; do
      if [ $lib = old ]; then export BANET_HIP_LIB=$OLD; else unset BANET_HIP_LIB; fi
      r=$(env $cfg PN=${PN:-5} PBITS=0 PROUNDS=2 timeout 300 python tools/prof_assemble.py 2>/dev/null | grep "us/window" | sed 's/.*kernels (best of 2): //')
      echo "rep $rep [$cfg] $lib $r"
    done
  done
done | tee $OUT/abq.log
exit 0
