#!/bin/bash
# Round 4: 8-row strip segments on mid-size two-frame launches: parity, then forced 8-row / 16-row / no strip over the launches in question.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -k "strip" ) > $OUT/k_pytest.log 2>&1
tail -4 $OUT/k_pytest.log
: > $OUT/k_sweep.txt
for cfg in "32 120 160" "8 240 320" "16 240 320" "2 480 640" "4 480 640" "64 120 160" "16 120 160"; do
  set -- $cfg
  PB=$1 PH=$2 PW=$3 PBITS=0,524288,263168,262144 PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/k_sweep.txt 2>&1
done
grep "us/launch" $OUT/k_sweep.txt
exit 0
