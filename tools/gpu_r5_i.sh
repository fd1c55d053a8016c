#!/bin/bash
# Round 5: pose + intrinsics as 16 scalars read once per (window, target frame) instead of vector loads + vmcnt(0) round trips inside
# strip_geometry (lib) against the previous build (lib_base): parity of the strip / quad kernels, then same-box alternating timing.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -q --timeout 800 -p no:cacheprovider -rfE --tb=short -k "strip or quad or mask_output or cfg3 or cfg5 or batch_invariant" ) > $OUT/r5i_pytest.log 2>&1
tail -n 6 $OUT/r5i_pytest.log | cut -c1-300
export PROUNDS=2 PN=4 PBITS=0
run() {  # label, env...
  local label=$1; shift
  for lib in lib_base lib lib_base lib; do
    echo "== $lib $label" | tee -a $OUT/r5i_ab.txt
    env BANET_HIP_LIB=$PWD/banet_amd/$lib/libbanet_hip.so "$@" timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5i_ab.txt
  done
}
run "640x480 x32 4 target frames" PB=32 PP=4
run "640x480 x32 2-frame" PB=32 PP=1
run "160x120 x32 2-frame (8-row strips)" PB=32 PP=1 PH=120 PW=160
run "80x60 x32 2-frame (quad)" PB=32 PP=1 PH=60 PW=80 PN=20
run "640x480 x1 2-frame (quad)" PB=1 PP=1 PN=20
exit 0
