#!/bin/bash
# Round 4, last call: FETCH_SIZE / WRITE_SIZE of a dense training step (forward + fused backward, 32 windows) on the final build, and
# bench.py exactly as the driver runs it (no arguments) -> the compact line.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_bwd_$c
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "adj_|target_map_adjoint|ba_gather|ba_syrk" --output-format csv -d /tmp/pmc_bwd_$c -o p -- python $REPO/tools/bench_dense_train.py 32 480 640 2 > $REPO/$OUT/last_pmc_bwd_$c.log 2>&1)
  echo "bwd $c exit $?"
done
{ python tools/summarize_pmc.py /tmp/pmc_bwd_FETCH_SIZE; python tools/summarize_pmc.py /tmp/pmc_bwd_WRITE_SIZE; } > $OUT/last_pmc_backward_fetch_write.txt 2>&1
head -30 $OUT/last_pmc_backward_fetch_write.txt | cut -c1-200
( time timeout 1500 python bench.py ) > $OUT/last_bench_default.log 2> $OUT/last_bench_default.err; echo "bench exit $?" >> $OUT/last_bench_default.err
tail -4 $OUT/last_bench_default.err
python - <<'PY'
import json
last = open("gpurun_out/last_bench_default.log").read().strip().split("\n")[-1]
d = json.loads(last)
print("compact line bytes", len(last), d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["parity"])
print({k: (v["value"], v.get("traffic_x"), v["parity_ok"], v["mask_flips"]) for k, v in d["sweep"].items()})
PY
exit 0
