#!/bin/bash
# phase timeline of the patch gather from the -DBANET_TIMING build (banet_amd/lib_timing, see tools/time_gather.py)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
export BANET_HIP_LIB=$PWD/banet_amd/lib_timing/libbanet_hip.so
for pb in ${PBS:-32 8}; do
  echo "== windows $pb" | tee -a $OUT/time_gather.log
  PB=$pb timeout 300 python tools/time_gather.py 2>&1 | grep -v amdgpu | tee -a $OUT/time_gather.log
done
exit 0
