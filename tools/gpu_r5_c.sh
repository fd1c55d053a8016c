#!/bin/bash
# Round 5, third GPU call: (1) the corrected round-5 tests; (2) is the strip gather paced by memory?  -DBANET_ABLATE_STRIP builds: the
# same instruction stream with the window rows (=1) / window rows + source rows + basis rows (=2) read from a cache-resident
# footprint, against the product build, at 32 windows x 4 target frames and x 1, same box, alternating.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q --timeout 800 -p no:cacheprovider -rfE --tb=short ) > $OUT/r5c_pytest_round5.log 2>&1
tail -n 12 $OUT/r5c_pytest_round5.log | cut -c1-300
export PB=32 PROUNDS=2 PN=4 PBITS=0
for pp in 4 1; do
  for lib in lib lib_ablate1 lib_ablate2 lib; do
    echo "== $lib pairs=$pp" | tee -a $OUT/r5c_ablate.txt
    BANET_HIP_LIB=$PWD/banet_amd/$lib/libbanet_hip.so PP=$pp timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5c_ablate.txt
  done
done
exit 0
