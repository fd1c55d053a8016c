#!/bin/bash
# unit-order experiment of the patch gather: time and L2 / fabric request counts for two reserved_ settings
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
BITS=${BITS:-"0 131072"}
PB=32 PBITS=$(echo $BITS | tr ' ' ',') PROUNDS=3 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/order_a.log
REPO=$PWD
for bits in $BITS; do
  rm -rf /tmp/pmc_o$bits
  (cd /tmp && PB=32 PBITS=$bits PROUNDS=1 timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_o$bits -o p -- python $REPO/tools/prof_assemble.py > /dev/null 2>&1)
  echo "bits $bits"; python tools/summarize_pmc.py /tmp/pmc_o$bits 2>/dev/null | grep -E "gather" | cut -c1-160
done | tee $OUT/order_pmc.log
exit 0
