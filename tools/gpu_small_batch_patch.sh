#!/bin/bash
# one window: direct gather (default below 4 tiles per resident wave) vs the patch kernel forced (reserved_ bit 9)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for cfg in "PB=1" "PB=1 PH=240 PW=320" "PB=2 PH=240 PW=320" "PB=4 PH=240 PW=320" "PB=1 PH=120 PW=160"; do
  env $cfg PBITS=0,512 PROUNDS=3 PN=3 timeout 200 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | grep "us/window" | cut -c1-200
done | tee $OUT/small_batch_patch.log
exit 0
