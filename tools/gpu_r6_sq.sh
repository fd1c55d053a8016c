#!/bin/bash
# Round 6, last: SQ counters of the backward's two large kernels on the final build (the recipe of tools/gpu_r5_a.sh; counters only with
# --kernel-trace): VALU pipe busy, what a wave does with its life -- the measured form of DESIGN 4.8's "bound by instruction execution".
set -u
OUT=gpurun_out/r6sq; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
python -c "from banet_amd import _capi; print('build id', _capi.lib().banet_build_id().decode())" 2>/dev/null | tail -1 | tee $OUT/sq.txt
run() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "adj_pixel2|adj_tile2|adj_basis6" --output-format csv -d /tmp/pmc_$name -o p -- python $REPO/tools/bench_dense_train.py 32 480 640 2 > $REPO/$OUT/pmc_$name.log 2>&1)
  echo "== [$*] exit $?" >> $OUT/sq.txt
  python tools/summarize_pmc.py /tmp/pmc_$name 2>/dev/null | grep -E "adj_" >> $OUT/sq.txt
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
run sq2 GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD
run sq3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 SQ_INSTS_VALU_FMA_F32 SQ_VALU_MFMA_BUSY_CYCLES
cat $OUT/sq.txt | cut -c1-200
