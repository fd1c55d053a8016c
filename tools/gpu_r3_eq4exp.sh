#!/bin/bash
# where does a step of eq_syrk4_kernel go?  EXP=1: no MFMAs (split + LDS + barrier), EXP=2: no split (LDS + MFMA + barrier)
OUT=gpurun_out/r3_eq4; mkdir -p $OUT
export EQ_SHAPES=8x76800x262
for m in 0 1 2; do
  lib=$PWD/banet_amd/lib/libbanet_hip.so; [ $m != 0 ] && lib=$PWD/banet_amd/lib_eq$m/libbanet_hip.so
  echo "== EXP=$m" | tee -a $OUT/exp.log
  BANET_HIP_LIB=$lib timeout 300 python tools/bench_eqcon.py 2>&1 | grep "^B=" | tee -a $OUT/exp.log
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eq4prof -o eq -- python $GRAFT_REPO_ROOT/tools/bench_eqcon.py > /dev/null 2>&1   # (without the timeout this line once ran into the call's limit)
cd $GRAFT_REPO_ROOT; f=$(ls $OUT/prof/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-200 | tee -a $OUT/exp.log
