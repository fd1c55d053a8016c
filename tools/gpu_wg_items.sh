#!/bin/bash
# NOTE: the reserved_ bit this script toggles belongs to an experiment that is no longer in the library (its code: see
# banet_amd/csrc/experiments/README.md and the *.patch.txt / *.hip.txt files there); kept as the record of how the numbers were taken.
# workgroup items (reserved_ bit 22) vs per-wave items: time, bit-identity, then fabric requests (PMC)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
PB=8 PBITS=${WBITS:-0,4194304} PROUNDS=2 timeout 90 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee $OUT/wg_items.log
[ ${PIPESTATUS[0]} -eq 0 ] || { echo "first run failed or timed out: stop"; exit 0; }
PB=32 PBITS=${WBITS:-0,4194304} PROUNDS=3 timeout 120 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee -a $OUT/wg_items.log
REPO=$PWD
for bits in ${PMCBITS:-0 4194304}; do
  rm -rf /tmp/pmc_o$bits
  (cd /tmp && PB=32 PBITS=$bits PROUNDS=1 timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_o$bits -o p -- python $REPO/tools/prof_assemble.py > /dev/null 2>&1)
  echo "bits $bits"; python tools/summarize_pmc.py /tmp/pmc_o$bits 2>/dev/null | grep -E "gather" | cut -c1-160
done | tee -a $OUT/wg_items.log
exit 0
