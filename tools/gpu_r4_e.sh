#!/bin/bash
# Round 4, fifth GPU call: the quad gather's two forms (one wave per SIMD with all loads in flight / two waves per SIMD) forced over
# level sizes x batches, then B = 1 / 8 bench lines.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -k "quad or cg_solve or mask or f16" ) > $OUT/e_pytest.log 2>&1
tail -4 $OUT/e_pytest.log
: > $OUT/e_sweep.txt
Q=33554432; D=$((Q+128)); S=$((Q+8192)); N=1073741824
for cfg in "1 30 40" "1 60 80" "1 120 160" "1 240 320" "1 480 640" "8 30 40" "8 60 80" "8 120 160" "8 240 320" "32 30 40" "32 60 80" "32 120 160"; do
  set -- $cfg
  PB=$1 PH=$2 PW=$3 PBITS=$D,$S,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/e_sweep.txt 2>&1
done
PB=32 PH=60 PW=80 PP=4 PBITS=$D,$S,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/e_sweep.txt 2>&1
PB=32 PH=30 PW=40 PP=4 PBITS=$D,$S,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/e_sweep.txt 2>&1
PB=1 PH=120 PW=160 PK=32 PBITS=$D,$S,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/e_sweep.txt 2>&1
grep "us/launch" $OUT/e_sweep.txt
exit 0
