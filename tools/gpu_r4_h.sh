#!/bin/bash
# Round 4: whole GPU suite on the current build (reduce2 split, column maxima from the first iteration's SYRK), headline A/B vs no fp16 SYRK, B = 1.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=5 ) > $OUT/h_pytest.log 2>&1
tail -12 $OUT/h_pytest.log
show() {
python - "$@" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "NO JSON LINE"); sys.exit(0)
d = json.loads(l[-1]); r = d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], "frac", r["frac"], "gather", r["per_level_gather_us"], "syrk", r["syrk_kernel"]["avg_launch_us"], "parity", d.get("parity"))
PY
}
for bits in 0 -2147483648 0; do
  timeout 900 python bench.py --steps 8 --warmup 3 --no-sweep --no-cpu-baseline --reserved=$bits > $OUT/h_bench_32_$bits.log 2> $OUT/h_bench_32_$bits.err
  show $OUT/h_bench_32_$bits.log "B32 bits $bits"
done
python - <<'PY'
import json
d = json.load(open("bench_detail.json"))
print("levels", {k: v.get("level_ms_last_step") for k, v in d["roofline"]["per_level"].items()}, d["ms_per_step"])
PY
timeout 600 python bench.py --windows 1 --steps 5 --warmup 2 --no-sweep --no-parity --no-cpu-baseline > $OUT/h_bench_1.log 2>&1
show $OUT/h_bench_1.log "B1"
exit 0
