#!/bin/bash
# Round 4: where does a batch-1 iteration go?  rocprofv3 kernel stats of `bench.py --windows 1`, + bench lines B = 1 / 8.
set -u
OUT=gpurun_out
mkdir -p $OUT $OUT/prof_b1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b1 -- python $REPO/bench.py --windows 1 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-sweep > $REPO/$OUT/g_prof_run.log 2>&1)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" $OUT/prof_b1/; done
head -14 $OUT/prof_b1/*kernel_stats.csv | cut -c1-200
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
for B in 1 8; do
  timeout 600 python bench.py --windows $B --steps 5 --warmup 2 --no-sweep --no-parity --no-cpu-baseline > $OUT/g_bench_$B.log 2>&1
  show $OUT/g_bench_$B.log "B$B"
done
timeout 300 python bench.py --windows 1 --height 120 --width 160 --basis 32 --iters 3 --steps 20 --warmup 3 --no-sweep --no-parity --no-cpu-baseline > $OUT/g_bench_cfg1like.log 2>&1
show $OUT/g_bench_cfg1like.log "160x120-K32-5level"
exit 0
