#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
PB=8 PBITS=0,16384 PROUNDS=3 timeout 600 python tools/prof_assemble.py > $OUT/u4_a.log 2>&1
cut -c1-200 $OUT/u4_a.log
PB=8 PH=240 PW=320 PBITS=0,16384 PROUNDS=3 timeout 600 python tools/prof_assemble.py > $OUT/u4_b.log 2>&1
cut -c1-200 $OUT/u4_b.log
PB=32 PBITS=0,16384 PROUNDS=2 timeout 600 python tools/prof_assemble.py > $OUT/u4_c.log 2>&1
cut -c1-200 $OUT/u4_c.log
exit 0
