#!/bin/bash
# Round 4, first GPU call: the frame-parallel strip gather (multi-frame windows).  Parity tests of the strip kernel, cfg-3 A/B
# (FP default / frames looped inside a wave = bit 22 / FP + slice barriers = bit 24), FETCH_SIZE passes of both.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -x -rfE --tb=short -k "strip or cfg3 or cfg5" ) > $OUT/fp_pytest.log 2>&1
tail -5 $OUT/fp_pytest.log
for bits in 0 4194304 16777216 0 4194304; do
  timeout 600 python bench.py --frames 5 --steps 3 --warmup 1 --no-sweep --no-parity --no-cpu-baseline --reserved $bits > $OUT/fp_cfg3_$bits.log 2>&1
  python - $OUT/fp_cfg3_$bits.log $bits <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
d = json.loads(l[-1]); r = d["roofline"]
print("cfg3 bits", sys.argv[2], d["value"], d["ms_per_step"], "frac", r["frac"], r["per_level_gather_us"], r["kernel"])
PY
done
for bits in 0 4194304; do
  rm -rf /tmp/pmc_$bits
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$bits -o p -- python $REPO/bench.py --frames 5 --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline --reserved $bits > $REPO/$OUT/fp_pmc_$bits.log 2>&1)
  echo "pmc $bits exit $?"
  python tools/summarize_pmc.py /tmp/pmc_$bits > $OUT/fp_pmc_fetch_$bits.txt 2>&1
  grep -i "gather" $OUT/fp_pmc_fetch_$bits.txt | head -8
done
# cfg-5 share quick A/B
for bits in 0 4194304; do
  timeout 900 python bench.py --frames 8 --height 960 --width 1280 --basis 256 --iters 15 --windows 8 --steps 2 --warmup 1 --no-sweep --no-parity --no-cpu-baseline --reserved $bits > $OUT/fp_cfg5_$bits.log 2>&1
  python - $OUT/fp_cfg5_$bits.log $bits <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
d = json.loads(l[-1]); r = d["roofline"]
print("cfg5 bits", sys.argv[2], d["value"], d["ms_per_step"], "frac", r["frac"], r["per_level_gather_us"], r["kernel"])
PY
done
exit 0
