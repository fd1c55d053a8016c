#!/bin/bash
# Round 5, fourth GPU call: (1) VALU issue rates on gfx950 as the gather sees them (tools/probe/valu_rate.hip); (2) the strip gather
# with UNPACKED channel maths (-DBANET_TAP_SCALAR -fno-slp-vectorize: v_fma_f32 instead of v_pk_fma_f32) against the product build;
# (3) the batch-invariance test with its corrected tolerance.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 120 tools/probe/valu_rate 2>&1 | tee $OUT/r5d_valu_rate.txt
( BANET_HIP_LIB=$PWD/banet_amd/lib_scalar/libbanet_hip.so timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q --timeout 500 -p no:cacheprovider -x -rfE --tb=short -k "strip or mask_output or cfg3 or quad" ) > $OUT/r5d_pytest_scalar.log 2>&1
echo "== strip + quad parity tests on the unpacked build: $(tail -n 1 $OUT/r5d_pytest_scalar.log)"
export PB=32 PROUNDS=2 PN=4 PBITS=0
for pp in 4 1; do
  for lib in lib lib_scalar lib lib_scalar; do
    echo "== $lib pairs=$pp" | tee -a $OUT/r5d_scalar.txt
    BANET_HIP_LIB=$PWD/banet_amd/$lib/libbanet_hip.so PP=$pp timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5d_scalar.txt
  done
done
( timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q --timeout 800 -p no:cacheprovider -rfE --tb=short -k batch_invariant ) > $OUT/r5d_pytest_round5.log 2>&1
tail -n 3 $OUT/r5d_pytest_round5.log | cut -c1-300
exit 0
