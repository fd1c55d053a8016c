#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel stats (+ optional PMC pass).
# Usage (on the GPU box, from the repo root):  bash tools/gpu_round.sh [tests|bench|prof|pmc|all]
set -u
MODE=${1:-all}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
if [[ $MODE == tests || $MODE == all ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -rfE --tb=short > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -40 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  tail -3 $OUT/smoke.log
fi
if [[ $MODE == bench || $MODE == all ]]; then
  timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $OUT/bench.log 2>&1; echo "bench exit $?" >> $OUT/bench.log
  tail -5 $OUT/bench.log
fi
if [[ $MODE == prof || $MODE == all ]]; then
  rm -rf /tmp/prof && mkdir -p /tmp/prof $OUT/prof
  REPO=$PWD
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_run.log 2>&1)
  echo "prof exit $?" >> $OUT/prof_run.log
  tail -3 $OUT/prof_run.log
  find /tmp/prof -type f | head -20
  for f in $(find /tmp/prof -name "*stats*.csv"); do cp "$f" $OUT/prof/; done
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f"
fi
if [[ $MODE == pmc || $MODE == all ]]; then
  rm -rf /tmp/pmc && mkdir -p /tmp/pmc
  REPO=$PWD
  for CNT in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d /tmp/pmc -o $CNT -- python $REPO/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline > $REPO/$OUT/pmc_run_$CNT.log 2>&1)
    echo "pmc $CNT exit $?" >> $OUT/pmc_run_$CNT.log
    tail -2 $OUT/pmc_run_$CNT.log
  done
  find /tmp/pmc -type f | head
  python tools/summarize_pmc.py /tmp/pmc > $OUT/pmc_summary.txt 2>&1
  tail -30 $OUT/pmc_summary.txt
fi
exit 0
