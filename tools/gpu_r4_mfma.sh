#!/bin/bash
# Round 4: matrix-pipe counters of the SYRK's two forms (exact bf16 x 3 pieces / 6 products vs fp16 x 2 pieces / 3 products) over one
# 640x480 x 32 assembly pass (the fp16 form forced into the single pass with reserved_ bit 24): MFMA busy cycles, MFMA instructions.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
export PB=32 PROUNDS=1 PN=4
for bits in 0 16777216; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
    name=$(echo $bits $set | tr ' ' '_' | cut -c1-40)
    rm -rf /tmp/pmc_$name
    (cd /tmp && PBITS=$bits timeout 600 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "ba_syrk|ba_recmax|ba_colmax" --output-format csv -d /tmp/pmc_$name -o p -- python $REPO/tools/prof_assemble.py > $REPO/$OUT/mfma_$name.log 2>&1)
    echo "== bits $bits [$set] exit $?"
    python tools/summarize_pmc.py /tmp/pmc_$name 2>/dev/null | grep -E "syrk|recmax|colmax"
  done
done
exit 0
