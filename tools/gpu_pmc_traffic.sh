#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline > $REPO/$OUT/pmc_$c.log 2>&1)
  echo "$c exit $?"
done
python tools/make_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 32 $OUT/pmc_traffic.json
{ python tools/summarize_pmc.py /tmp/pmc_FETCH_SIZE; python tools/summarize_pmc.py /tmp/pmc_WRITE_SIZE; } > $OUT/pmc_fetch_write.txt 2>&1
exit 0
