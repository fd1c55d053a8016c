#!/bin/bash
# Round 4: quad gather with static item assignment (no atomic queue) forced / off over level sizes x batches.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -k "quad or cg_solve or mask or f16" ) > $OUT/f_pytest.log 2>&1
tail -4 $OUT/f_pytest.log
: > $OUT/f_sweep.txt
Q=33554432; N=1073741824
for cfg in "1 30 40" "1 60 80" "1 120 160" "1 240 320" "1 480 640" "8 30 40" "8 60 80" "8 120 160" "8 240 320" "32 30 40" "32 60 80" "32 120 160"; do
  set -- $cfg
  PB=$1 PH=$2 PW=$3 PBITS=$Q,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/f_sweep.txt 2>&1
done
PB=32 PH=60 PW=80 PP=4 PBITS=$Q,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/f_sweep.txt 2>&1
PB=32 PH=30 PW=40 PP=4 PBITS=$Q,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/f_sweep.txt 2>&1
PB=1 PH=120 PW=160 PK=32 PBITS=$Q,$N PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/f_sweep.txt 2>&1
grep "us/launch" $OUT/f_sweep.txt
show() {
python - "$@" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "NO JSON LINE"); sys.exit(0)
d = json.loads(l[-1]); r = d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], "frac", r["frac"], "gather", r["per_level_gather_us"], "syrk", r["syrk_kernel"]["avg_launch_us"])
PY
}
for B in 1 8 32; do
  timeout 600 python bench.py --windows $B --steps 5 --warmup 2 --no-sweep --no-parity --no-cpu-baseline > $OUT/f_bench_$B.log 2>&1
  show $OUT/f_bench_$B.log "B$B"
done
exit 0
