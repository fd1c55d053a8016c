#!/bin/bash
# NOTE: the reserved_ bit this script toggles belongs to an experiment that is no longer in the library (its code: see
# banet_amd/csrc/experiments/README.md and the *.patch.txt / *.hip.txt files there); kept as the record of how the numbers were taken.
# tile-mode patch gather (default) vs per-unit patches (reserved_ bit 21): parity tests, then same-process A/B
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) 2>&1 | tail -12 | tee $OUT/tile_tests.log
for cfg in "PB=32" "PB=8" "PB=32 PH=240 PW=320" "PB=8 PP=4"; do
  env $cfg PBITS=0,2097152 PROUNDS=3 timeout 300 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | cut -c1-230
done | tee $OUT/tile_ab.log
exit 0
