#!/bin/bash
# Round 4, fourth GPU call: the whole GPU suite on the build with the quad gather + fp16 SYRK defaults, then same-box A/B of the
# headline: default / no fp16 SYRK (bit 31) / no quad gather (bit 30) / neither.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=5 ) > $OUT/d_pytest.log 2>&1
tail -15 $OUT/d_pytest.log
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
for bits in 0 -2147483648 1073741824 -1073741824 0; do
  timeout 900 python bench.py --steps 8 --warmup 3 --no-sweep --no-cpu-baseline --reserved=$bits > $OUT/d_bench_32_$bits.log 2> $OUT/d_bench_32_$bits.err
  show $OUT/d_bench_32_$bits.log "B32 bits $bits"
  tail -2 $OUT/d_bench_32_$bits.err | cut -c1-300
done
timeout 900 python bench.py --frames 5 --steps 4 --warmup 1 --no-sweep --no-parity --no-cpu-baseline > $OUT/d_bench_cfg3.log 2>&1
show $OUT/d_bench_cfg3.log "cfg3"
exit 0
