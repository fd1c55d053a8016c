#!/bin/bash
# quick A/B of the patch gather variants (prof_assemble.py) + the forward parity tests
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
PB=8 PBITS=0,65536 PROUNDS=3 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/qg_a.log
PB=8 PH=240 PW=320 PBITS=0,65536 PROUNDS=3 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/qg_b.log
PB=32 PBITS=0,65536 PROUNDS=2 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/qg_c.log
PB=8 PP=4 PBITS=0,65536 PROUNDS=2 timeout 600 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee $OUT/qg_d.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) 2>&1 | tail -6
exit 0
