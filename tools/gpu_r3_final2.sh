#!/bin/bash
# last call of round 3: GPU tests + smoke, the PMC traffic passes for this build id, the default bench line (short), the opt-in
# three-product SYRK beside it (bench.py --reserved 536870912), each with its in-line parity
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
( timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3 ) | tee $OUT/pytest_gpu_short.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline > $REPO/$OUT/pmc_$c.log 2>&1)
  echo "$c exit $?"
done
python tools/make_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 32 $OUT/pmc_traffic.json | cut -c1-200
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
for bits in 0 536870912; do
  timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --reserved $bits > $OUT/bench_r$bits.log 2>/dev/null
  python - <<PY
import json
l=[x for x in open("$OUT/bench_r$bits.log") if x.startswith("{")]
d=json.loads(l[0]); r=d["roofline"]; p=d["parity"]
print("reserved $bits:", d["value"], d["ms_per_step"], "gather frac", r["frac"], "traffic", r["traffic"], "syrk", r["syrk_kernel"]["avg_launch_us"], "parity", p["ok"], p["max_rel_err"])
PY
done
