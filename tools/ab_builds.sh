#!/bin/bash
# Same-box A/B of two builds of the library: alternates `prof_assemble.py` between the in-tree build and
# banet_amd/lib_ab/libbanet_hip_old.so (BANET_HIP_LIB override of banet_amd/_capi.py).  usage: bash tools/ab_builds.sh
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
OLD=$PWD/banet_amd/lib_ab/libbanet_hip_old.so
for rep in 1 2 3; do
  for cfg in "8 480 640 1" "32 480 640 1" "8 240 320 1" "8 480 640 4"; do
    set -- $cfg
    for lib in new old; do
      if [ $lib = old ]; then export BANET_HIP_LIB=$OLD; else unset BANET_HIP_LIB; fi
      r=$(PB=$1 PH=$2 PW=$3 PP=$4 PBITS=0 PROUNDS=2 timeout 300 python tools/prof_assemble.py 2>/dev/null | grep "us/window" | sed 's/.*kernels (best of 2): //')
      echo "rep $rep B=$1 ${3}x$2 pairs=$4 $lib $r"
    done
  done
done | tee $OUT/ab_builds.log
exit 0
