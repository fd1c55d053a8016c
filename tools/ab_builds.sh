#!/bin/bash
# Same-box A/B of two builds of the library: alternates `prof_assemble.py` between the in-tree build and
# banet_amd/lib_ab/libbanet_hip_old.so (BANET_HIP_LIB override of banet_amd/_capi.py); PN = launches per timed burst
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
OLD=$PWD/banet_amd/lib_ab/libbanet_hip_old.so
for rep in 1 2; do
  for pn in 5 60; do
    for lib in new old; do
      if [ $lib = old ]; then export BANET_HIP_LIB=$OLD; else unset BANET_HIP_LIB; fi
      r=$(PN=$pn PB=32 PBITS=0 PROUNDS=2 timeout 300 python tools/prof_assemble.py 2>/dev/null | grep "us/window" | sed 's/.*kernels (best of 2): //')
      echo "rep $rep B=32 640x480 burst=$pn $lib $r"
    done
  done
done | tee $OUT/ab_builds.log
exit 0
