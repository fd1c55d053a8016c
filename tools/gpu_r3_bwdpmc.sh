#!/bin/bash
# fabric traffic of the backward's kernels: FETCH_SIZE / WRITE_SIZE passes over one 32-window training step
set -u
OUT=gpurun_out/r3_bwdpmc; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bp_$c
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/bp_$c -o p -- python $REPO/tools/bench_dense_train.py 32 480 640 2 > $REPO/$OUT/run_$c.log 2>&1)
  python tools/summarize_pmc.py /tmp/bp_$c 2>&1 | grep -i "adj_\|target_map\|gather128s\|syrk\|Kernel\|kernel" | head -14 | tee -a $OUT/summary.txt
done
