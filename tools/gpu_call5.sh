#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
PB=8 PBITS=0,1,2,3,7,15,9,5,4,8 PROUNDS=2 timeout 600 python tools/prof_assemble.py > $OUT/ablate_p.log 2>&1
cat $OUT/ablate_p.log | grep -v "^  " | cut -c1-200
exit 0
