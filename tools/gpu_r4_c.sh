#!/bin/bash
# Round 4, third GPU call: the 4x4-pixel-item gather (ba_gather128q_kernel) and the fp16 two-piece SYRK -- parity tests, the quad
# gather forced on / off over level sizes x batches (where does it pay?), bench lines with / without either.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -s -k "quad or mask or twin or f16 or equation_construction" ) > $OUT/c_pytest.log 2>&1
grep -E "fp16 two-piece|passed|failed|FAILED|Error" $OUT/c_pytest.log | tail -30
: > $OUT/c_sweep.txt
for cfg in "1 30 40" "1 60 80" "1 120 160" "1 240 320" "1 480 640" "8 30 40" "8 60 80" "8 120 160" "8 240 320" "32 30 40" "32 60 80" "32 120 160"; do
  set -- $cfg
  PB=$1 PH=$2 PW=$3 PBITS=33554432,1073741824 PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/c_sweep.txt 2>&1
done
PB=32 PH=60 PW=80 PP=4 PBITS=33554432,1073741824 PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/c_sweep.txt 2>&1
PB=32 PH=30 PW=40 PP=4 PBITS=33554432,1073741824 PN=10 timeout 300 python tools/prof_assemble.py >> $OUT/c_sweep.txt 2>&1
grep "us/launch\|max rel diff" $OUT/c_sweep.txt
# fp16 SYRK at the kernel level: 640x480 x 8 (single pass, forced) vs the exact form
PB=8 PH=480 PW=640 PBITS=0,16777216 PN=10 timeout 300 python tools/prof_assemble.py 2>&1 | grep "us/launch\|max rel diff"
show() {
python - "$@" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
d = json.loads(l[-1]); r = d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], "frac", r["frac"], "gather", r["per_level_gather_us"], "syrk", r["syrk_kernel"], "parity", d.get("parity"))
PY
}
for bits in 0 -2147483648 1073741824; do
  timeout 900 python bench.py --steps 6 --warmup 2 --no-sweep --no-cpu-baseline --reserved=$bits > $OUT/c_bench_32_$bits.log 2>&1
  show $OUT/c_bench_32_$bits.log "B32 bits $bits"
done
for B in 1 8; do
  for bits in 0 1073741824; do
    timeout 600 python bench.py --windows $B --steps 5 --warmup 2 --no-sweep --no-parity --no-cpu-baseline --reserved=$bits > $OUT/c_bench_${B}_$bits.log 2>&1
    show $OUT/c_bench_${B}_$bits.log "B$B bits $bits"
  done
done
exit 0
