#!/bin/bash
# fixed-stride patch gather (buffer loads) for K = 0 (pose only) and K = 256: A/B against the packed patch (reserved_ bit 16) + parity
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
PB=8 PK=0 PBITS=0,65536 PROUNDS=3 timeout 300 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee $OUT/fsk_a.log
PB=8 PK=256 PBITS=0,65536 PROUNDS=3 timeout 300 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee $OUT/fsk_b.log
PB=4 PK=256 PP=4 PBITS=0,65536 PROUNDS=2 timeout 300 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-220 | tee $OUT/fsk_c.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) 2>&1 | tail -6 | tee $OUT/fsk_tests.log
exit 0
