#!/bin/bash
# Round 5, second GPU call: (1) the new -m gpu tests (production selection of cfg-3 / cfg-5 at full size, batch invariance);
# (2) the strip gather's OPT variants (BANET_STRIP_OPT: 1 = both pieces' window reads before the maths, 2 = one M0 write per row,
# 3 = both): parity tests of the strip kernel under each, then same-box alternating timing at 32 windows x 4 target frames and
# x 1; (3) in-kernel timelines of the frame-parallel kernel (-DBANET_TIMING=3/4/5 builds).
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q --timeout 800 -p no:cacheprovider -rfE --tb=short ) > $OUT/r5b_pytest_round5.log 2>&1
tail -n 15 $OUT/r5b_pytest_round5.log
for opt in 3 1 2; do
  ( BANET_STRIP_OPT=$opt timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q --timeout 500 -p no:cacheprovider -x -rfE --tb=short -k "strip or mask_output or cfg3" ) > $OUT/r5b_pytest_opt$opt.log 2>&1
  echo "== strip parity tests under BANET_STRIP_OPT=$opt: $(tail -n 1 $OUT/r5b_pytest_opt$opt.log)"
done
export PB=32 PROUNDS=2 PN=4 PBITS=0
for opt in 0 1 2 3 0 3; do
  echo "== BANET_STRIP_OPT=$opt" | tee -a $OUT/r5b_timing.txt
  BANET_STRIP_OPT=$opt PP=4 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5b_timing.txt
done
for opt in 0 3 0 3; do
  echo "== BANET_STRIP_OPT=$opt" | tee -a $OUT/r5b_timing.txt
  BANET_STRIP_OPT=$opt PP=1 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5b_timing.txt
done
for m in 3 4 5; do
  echo "== BANET_TIMING=$m pairs=4" | tee -a $OUT/r5b_time_strip.txt
  BANET_HIP_LIB=$PWD/banet_amd/lib_timing$m/libbanet_hip.so PMODE=$m PP=4 timeout 300 python tools/time_strip.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5b_time_strip.txt
done
for m in 4 5; do
  echo "== BANET_TIMING=$m pairs=1" | tee -a $OUT/r5b_time_strip.txt
  BANET_HIP_LIB=$PWD/banet_amd/lib_timing$m/libbanet_hip.so PMODE=$m PP=1 timeout 300 python tools/time_strip.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5b_time_strip.txt
done
exit 0
