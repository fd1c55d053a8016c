#!/bin/bash
# round 6: target-tile backward -- tests, training-step A/B, kernel stats of the fold mode
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gpu_round6.py -m gpu -q --timeout 900 -p no:cacheprovider -rfE --tb=short ) 2>&1 | tail -25 | tee $OUT/r6b_tests.txt
for tile in 0 9 10 11 12 13 1; do
  echo "BANET_ADJOINT_TILE=$tile"
  BANET_ADJOINT_TILE=$tile timeout 600 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2
done | tee $OUT/r6b_dense_train.txt
cd /tmp
for tile in 0 10 11; do
  rm -rf /tmp/prof_$tile
  BANET_ADJOINT_TILE=$tile timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tile -o p -- python $GRAFT_REPO_ROOT/tools/bench_dense_train.py 32 480 640 2 > /tmp/prof_$tile.log 2>&1
  f=$(find /tmp/prof_$tile -name "*kernel_stats.csv" | head -1)
  echo "== tile shape $tile"; grep "adj_tile\|adj_pixel\|adj_basis" "$f" | cut -c1-200
done 2>&1 | tee $GRAFT_REPO_ROOT/$OUT/r6b_stats.txt
exit 0
