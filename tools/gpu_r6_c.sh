#!/bin/bash
# round 6: the HIP small step -- tests, training-step timing (A/B against the torch small step), sparse training iteration
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gpu_round6.py -m gpu -q --timeout 900 -p no:cacheprovider -rfE --tb=short -k "small_step" ) 2>&1 | tail -25 | tee $OUT/r6c_tests.txt
for hip in 1 0; do
  for w in 32 8; do
    echo "BANET_SMALL_STEP_HIP=$hip windows=$w"
    BANET_SMALL_STEP_HIP=$hip timeout 600 python tools/bench_dense_train.py $w 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2
  done
  echo "BANET_SMALL_STEP_HIP=$hip sparse training iteration"
  BANET_SMALL_STEP_HIP=$hip timeout 600 python tools/train_graph_bench.py 2>&1 | grep -v amdgpu | tail -6
done | tee $OUT/r6c_timing.txt
exit 0
