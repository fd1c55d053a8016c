#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_dense_backward.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) > $OUT/pytest_bwd.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_bwd.log
grep -v "^$" $OUT/pytest_bwd.log | tail -20
timeout 600 python tools/bench_dense_train.py 8 480 640 2 > $OUT/dense_train.log 2>&1; tail -4 $OUT/dense_train.log
PB=8 PBITS=0,8192 PROUNDS=2 timeout 600 python tools/prof_assemble.py > $OUT/occ1.log 2>&1
grep -v "^  " $OUT/occ1.log | cut -c1-200
rm -rf /tmp/prof && mkdir -p /tmp/prof $OUT/prof_train
REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o tr -- python $REPO/tools/bench_dense_train.py 8 480 640 1 > $REPO/$OUT/prof_train_run.log 2>&1)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" $OUT/prof_train/; done
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "adj_|target_map_adj" "$f" | cut -c1-160
exit 0
