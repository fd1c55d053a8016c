#!/bin/bash
# Round 5, first GPU call: what bounds the frame-parallel strip gather (cfg-3's 640x480 launch)?  SQ counters of one assembly pass at
# 32 windows x 4 target frames next to the 2-frame launch: VALU busy, LDS busy, wait cycles, wave cycles -- per SIMD and kernel cycle.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
export PB=32 PROUNDS=2 PN=4 PBITS=0
for pp in 4 1; do
  PP=$pp timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5a_timing.txt
done
export PROUNDS=1
run() {
  local pp=$1 name=$2; shift 2
  rm -rf /tmp/pmc_$name
  (cd /tmp && PP=$pp timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "ba_gather128s" --output-format csv -d /tmp/pmc_$name -o p -- python $REPO/tools/prof_assemble.py > $REPO/$OUT/r5a_pmc_$name.log 2>&1)
  echo "== pairs $pp [$*] exit $?" | tee -a $OUT/r5a_pmc_summary.txt
  python tools/summarize_pmc.py /tmp/pmc_$name 2>/dev/null | grep -E "gather" | tee -a $OUT/r5a_pmc_summary.txt
}
for pp in 4 1; do
  run $pp sq1_$pp SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
  run $pp sq2_$pp GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD
  run $pp sq3_$pp SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 SQ_INSTS_VALU_FMA_F32 SQ_VALU_MFMA_BUSY_CYCLES
done
exit 0
