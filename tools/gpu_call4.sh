#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python tools/bench_dense_train.py 8 480 640 2 > $OUT/dense_train.log 2>&1; echo "exit $?" >> $OUT/dense_train.log
cat $OUT/dense_train.log | tail -8
rm -rf /tmp/prof && mkdir -p /tmp/prof $OUT/prof_train
REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o tr -- python $REPO/tools/bench_dense_train.py 8 480 640 1 > $REPO/$OUT/prof_train_run.log 2>&1)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" $OUT/prof_train/; done
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-200
exit 0
