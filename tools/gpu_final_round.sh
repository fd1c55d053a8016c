#!/bin/bash
# Final artefacts of a round: GPU tests, smoke, bench line, rocprofv3 kernel stats of the bench command, PMC traffic passes.
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
( time timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 ) > $OUT/bench.log 2>&1; echo "bench exit $?" >> $OUT/bench.log
tail -c 300 $OUT/bench.log
exit 0
