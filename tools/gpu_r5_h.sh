#!/bin/bash
# Round 5, GPU call: (1) round-5 tests incl. the fused sparse iteration against float64; (2) literal op backward at P = 298 on the
# matrix pipe: parity + timing; (3) frame-parallel strip gather with 8- / 4-row depth-dot batches (no scratch) vs the previous build.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q --timeout 800 -p no:cacheprovider -rfE --tb=short -k "round5 or fused or equation or eqcon or strip or cfg3 or cfg5 or mask_output" ) > $OUT/r5h_pytest.log 2>&1
tail -n 12 $OUT/r5h_pytest.log | cut -c1-300
EQ_SHAPES=8x76800x298,2x76800x298,8x4096x298,8x76800x262 timeout 600 python tools/bench_eqcon.py 2>&1 | grep -v amdgpu | tee $OUT/r5h_eqcon.txt
export PB=32 PROUNDS=2 PN=4 PBITS=0
for lib in lib_rb16 lib lib_rb16 lib; do
  echo "== $lib pairs=4" | tee -a $OUT/r5h_fp_rb.txt
  BANET_HIP_LIB=$PWD/banet_amd/$lib/libbanet_hip.so PP=4 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5h_fp_rb.txt
done
for lib in lib_rb16 lib; do
  echo "== $lib cfg5 level: 8 windows x 7 pairs 1280x960 K=256" | tee -a $OUT/r5h_fp_rb.txt
  BANET_HIP_LIB=$PWD/banet_amd/$lib/libbanet_hip.so PB=8 PP=7 PH=960 PW=1280 PK=256 PN=2 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5h_fp_rb.txt
done
exit 0
