#!/bin/bash
# Round 4, after the final artefacts: fuzz seeds on the final build (round-4 kernels included), backward fuzz, and a re-test of
# the small backward step as a captured graph at 32 windows (pathologically slow in round 2: still?).
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1500 python tools/fuzz_parity.py 21000 60 ) > $OUT/post_fuzz.log 2>&1; echo "fuzz exit $?" >> $OUT/post_fuzz.log
grep -E "failures|FAILED|exit" $OUT/post_fuzz.log | cut -c1-200 | tail -12
( timeout 900 python tools/fuzz_backward.py 500 40 ) > $OUT/post_fuzz_bwd.log 2>&1; echo "fuzz_bwd exit $?" >> $OUT/post_fuzz_bwd.log
tail -3 $OUT/post_fuzz_bwd.log | cut -c1-200
for g in 1 2; do
  BANET_TRAIN_GRAPH=$g timeout 600 python tools/bench_dense_train.py 32 480 640 2 > $OUT/post_train_graph$g.log 2>&1
  echo "BANET_TRAIN_GRAPH=$g"; grep "forward\|small" $OUT/post_train_graph$g.log | head -4
done
exit 0
