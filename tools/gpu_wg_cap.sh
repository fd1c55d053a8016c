#!/bin/bash
# NOTE: the reserved_ bit this script toggles belongs to an experiment that is no longer in the library (its code: see
# banet_amd/csrc/experiments/README.md and the *.patch.txt / *.hip.txt files there); kept as the record of how the numbers were taken.
# mid-size levels (direct gather below 4 tiles per resident wave): the cap on resident workgroups, 320 (default) vs 224 / 160 / 96
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for cfg in "PB=1" "PB=1 PH=240 PW=320" "PB=2 PH=240 PW=320" "PB=4 PH=240 PW=320" "PB=1 PH=120 PW=160" "PB=8 PH=120 PW=160" "PB=32 PH=120 PW=160" "PB=32 PH=60 PW=80"; do
  env $cfg PBITS=0,4194304,8388608,16777216 PROUNDS=3 PN=3 timeout 200 python tools/prof_assemble.py 2>&1 | grep -v amdgpu | grep "us/window" | cut -c1-200
done | tee $OUT/wg_cap.log
exit 0
