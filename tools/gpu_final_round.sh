#!/bin/bash
# Final artefacts of a round in one gpurun call: GPU tests, smoke, PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes,
# counters on our kernels only) of the headline AND of cfg-3 / the cfg-5 share -> profiles/pmc_traffic.json (tied to this build by
# banet_build_id), rocprofv3 kernel stats of the bench command, the full bench line (reads that traffic file), bench.py's own
# 2-rank launch on this one-GPU box.
set -u
OUT=gpurun_out
mkdir -p $OUT $OUT/prof
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=5 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
pmc_pass() {   # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$c
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "ba_gather|ba_syrk" --output-format csv -d /tmp/pmc_${name}_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline "$@" > $REPO/$OUT/pmc_${name}_$c.log 2>&1)
    echo "$name $c exit $?"
  done
  { python tools/summarize_pmc.py /tmp/pmc_${name}_FETCH_SIZE; python tools/summarize_pmc.py /tmp/pmc_${name}_WRITE_SIZE; } > $OUT/pmc_fetch_write_$name.txt 2>&1
}
pmc_pass headline
python tools/make_pmc_traffic.py /tmp/pmc_headline_FETCH_SIZE /tmp/pmc_headline_WRITE_SIZE 32 profiles/pmc_traffic.json > /dev/null
pmc_pass cfg3 --frames 5
python tools/make_pmc_traffic.py /tmp/pmc_cfg3_FETCH_SIZE /tmp/pmc_cfg3_WRITE_SIZE 32 profiles/pmc_traffic.json cfg3_5frame_B32 5 480 640 128 10 > /dev/null
pmc_pass cfg5 --frames 8 --height 960 --width 1280 --basis 256 --iters 15 --windows 8
python tools/make_pmc_traffic.py /tmp/pmc_cfg5_FETCH_SIZE /tmp/pmc_cfg5_WRITE_SIZE 8 profiles/pmc_traffic.json cfg5_8frame_1280x960_K256_B8 8 960 1280 256 15 > /dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python - <<'PY'
import json
d = json.load(open("profiles/pmc_traffic.json"))
print("traffic headline", d["hbm_bytes_per_launch"], d["overfetch"], "syrk fetch KB", d["syrk_fetch_kb_per_launch"])
for k, v in d.get("workloads", {}).items():
    print("traffic", k, v["hbm_bytes_per_launch"], v["overfetch"])
PY
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o fin -- python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sweep > $REPO/$OUT/prof_run.log 2>&1)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" $OUT/prof/; done
( time timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
tail -3 $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
( BANET_BENCH_DEVICE=0 BANET_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --windows 8 ) > $OUT/bench_2rank.log 2> $OUT/bench_2rank.err; echo "2rank exit $?" >> $OUT/bench_2rank.err
tail -1 $OUT/bench_2rank.err
python - <<'PY'
import json
last = [x for x in open("gpurun_out/bench.log") if x.startswith("{")][-1]
print("compact line bytes", len(last))
d = json.load(open("gpurun_out/bench_detail.json")); r = d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_note"])
print("levels", {k: v.get("level_ms_last_step") for k, v in r["per_level"].items()})
print("parity", d["parity"]["ok"], d["parity"]["max_rel_err"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["median_s"])
for k, v in d["sweep"].items():
    p = v.get("parity", {})
    print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], "traffic", v["roofline"].get("traffic"), p.get("ok"), p.get("max_rel_err"), "flips", p.get("mask_bits_differing"))
PY
exit 0
