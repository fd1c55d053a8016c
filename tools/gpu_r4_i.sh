#!/bin/bash
# Round 4: the fp16 two-piece form in the syrk_wide.hip jobs (K = 256 / more than 4 target frames): tests, cfg-5 share A/B.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -s -k "f16 or cfg5 or cfg3 or wide or 256" ) > $OUT/i_pytest.log 2>&1
grep -E "fp16 two-piece|passed|failed|FAILED|Error" $OUT/i_pytest.log | tail -24
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
for bits in 0 -2147483648; do
  timeout 900 python bench.py --frames 8 --height 960 --width 1280 --basis 256 --iters 15 --windows 8 --steps 2 --warmup 1 --no-sweep --no-parity --no-cpu-baseline --reserved=$bits > $OUT/i_cfg5_$bits.log 2>&1
  show $OUT/i_cfg5_$bits.log "cfg5 bits $bits"
done
exit 0
