#!/bin/bash
# round-2 GPU call 1: new tests, new bench line, B=256 feasibility, sparse workload, kernel stats
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
nproc > $OUT/nproc.txt
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=15 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -45 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
tail -3 $OUT/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 ) > $OUT/bench.log 2>&1; echo "bench exit $?" >> $OUT/bench.log
tail -c 3000 $OUT/bench.log
timeout 600 python tools/bench_sparse.py > $OUT/bench_sparse.log 2>&1; echo "sparse exit $?" >> $OUT/bench_sparse.log
tail -5 $OUT/bench_sparse.log
timeout 600 python tools/bench_sparse.py --lm > $OUT/bench_sparse_lm.log 2>&1; echo "sparse-lm exit $?" >> $OUT/bench_sparse_lm.log
tail -3 $OUT/bench_sparse_lm.log
rm -rf /tmp/prof && mkdir -p /tmp/prof $OUT/prof
REPO=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r02 -- python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sweep > $REPO/$OUT/prof_run.log 2>&1)
echo "prof exit $?" >> $OUT/prof_run.log
for f in $(find /tmp/prof -name "*stats*.csv"); do cp "$f" $OUT/prof/; done
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f"
exit 0
