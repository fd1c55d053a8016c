#!/bin/bash
# PMC passes (one counter set per pass, --kernel-trace only) over one L0 assembly at the headline batch: MFMA busy, GRBM, SQ, LDS, TCC
export PB=${PB:-32} PROUNDS=1 PBITS=0
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
run() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $REPO/tools/prof_assemble.py > $REPO/$OUT/pmc_$name.log 2>&1)
  echo "== $name exit $?"
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
for n in mfma grbm sq1 lds tcc; do python tools/summarize_pmc.py /tmp/pmc_$n 2>/dev/null | grep -E "gather|syrk"; done | tee $OUT/pmc_kernels_summary.txt
exit 0
