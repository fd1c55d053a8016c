#!/bin/bash
# Final artefacts of a round: GPU tests, smoke, PMC traffic passes (-> profiles/pmc_traffic.json, tied to this build by
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
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline > $REPO/$OUT/pmc_$c.log 2>&1)
  echo "$c exit $?"
done
python tools/make_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 32 profiles/pmc_traffic.json
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
{ python tools/summarize_pmc.py /tmp/pmc_FETCH_SIZE; python tools/summarize_pmc.py /tmp/pmc_WRITE_SIZE; } > $OUT/pmc_fetch_write.txt 2>&1
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o fin -- python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sweep > $REPO/$OUT/prof_run.log 2>&1)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" $OUT/prof/; done
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
tail -3 $OUT/bench.err
( BANET_BENCH_DEVICE=0 BANET_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --windows 8 ) > $OUT/bench_2rank.log 2> $OUT/bench_2rank.err; echo "2rank exit $?" >> $OUT/bench_2rank.err
tail -1 $OUT/bench_2rank.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench.log") if x.startswith("{")]
d=json.loads(l[0]); r=d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_note"])
print("parity", d["parity"]["ok"], d["parity"]["max_rel_err"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["median_s"])
for k,v in d["sweep"].items():
    print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["parity"]["ok"], v["parity"]["max_rel_err"])
PY
exit 0
