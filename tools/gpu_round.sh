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
  rm -rf $OUT/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o r01 -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_run.log 2>&1)
  echo "prof exit $?" >> $OUT/prof_run.log
  find $OUT/prof -name "*kernel_stats*" | head -3
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
  # the trace itself is large; keep only the stats
  find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
fi
if [[ $MODE == pmc || $MODE == all ]]; then
  rm -rf $OUT/pmc
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OLDPWD/$OUT/pmc -o fetch -- python $OLDPWD/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline > $OLDPWD/$OUT/pmc_run.log 2>&1)
  echo "pmc fetch exit $?" >> $OUT/pmc_run.log
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OLDPWD/$OUT/pmc -o write -- python $OLDPWD/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline >> $OLDPWD/$OUT/pmc_run.log 2>&1)
  echo "pmc write exit $?" >> $OUT/pmc_run.log
  python tools/summarize_pmc.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
  tail -20 $OUT/pmc_summary.txt
fi
exit 0
