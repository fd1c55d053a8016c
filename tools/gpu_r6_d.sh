#!/bin/bash
# round 6: backward tests + training-step timing + kernel stats (after the adj_basis6 transposition)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/test_gpu_dense_backward.py tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q --timeout 900 -p no:cacheprovider -rfE --tb=line ) 2>&1 | tail -8 | tee $OUT/r6d_tests.txt
for w in 32 8; do timeout 600 python tools/bench_dense_train.py $w 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2; done | tee $OUT/r6d_dense_train.txt
PFRAMES=5 timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2 | tee -a $OUT/r6d_dense_train.txt
cd /tmp; rm -rf /tmp/prof_d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -o p -- python $GRAFT_REPO_ROOT/tools/bench_dense_train.py 32 480 640 2 > /tmp/prof_d.log 2>&1
f=$(find /tmp/prof_d -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-70,100-190 | tee $GRAFT_REPO_ROOT/$OUT/r6d_stats.txt
cp "$f" $GRAFT_REPO_ROOT/$OUT/r6d_kernel_stats.csv
exit 0
