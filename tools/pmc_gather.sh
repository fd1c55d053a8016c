#!/bin/bash
# PMC passes over tools/prof_assemble.py (L0 640x480, 8 windows): where do the gather / syrk waves spend time?
export PB=${PB:-8} PROUNDS=1
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
run() { # name counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $REPO/tools/prof_assemble.py > $REPO/$OUT/pmc_$name.log 2>&1)
  echo "== $name exit $?"; tail -2 $OUT/pmc_$name.log | cut -c1-300
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run ta TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TD_TD_BUSY_sum
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS
for n in sq1 sq2 tcc grbm tcp ta mfma lds; do python tools/summarize_pmc.py /tmp/pmc_$n 2>/dev/null | grep -E "gather|syrk" ; done | tee $OUT/pmc_gather_summary.txt
grep -h "gather_kernel" /tmp/pmc_grbm/*kernel_trace.csv 2>/dev/null | head -3 | cut -c1-400
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|gpu)|TCP_|TA_|TCC_HIT|TCC_MISS" | head -60 > $OUT/counters_avail.txt
