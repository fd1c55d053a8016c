#!/bin/bash
# timing-only ablations of the backward's kernels (wrong results on purpose): where do adj_basis6 / adj_pixel2 spend their time?
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
for abl in 0 1 2 3 4 8 12; do
  rm -rf /tmp/prof_a$abl
  BANET_ADJOINT_ABLATE=$abl timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a$abl -o p -- python $GRAFT_REPO_ROOT/tools/bench_dense_train.py 32 480 640 2 > /tmp/prof_a$abl.log 2>&1
  f=$(find /tmp/prof_a$abl -name "*kernel_stats.csv" | head -1)
  echo "== ablate $abl"; grep "adj_tile\|adj_pixel\|adj_basis" "$f" | cut -c1-60,100-200
done 2>&1 | tee $GRAFT_REPO_ROOT/$OUT/r6_ablate.txt
exit 0
