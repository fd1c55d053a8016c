#!/bin/bash
# Round 6, late: BANET_ADJOINT_REUSE_DEPTH_SEED (multi-frame windows: frames 2.. of an iteration reuse z2 / zeta / e) -- tests + same-box A/B
set -u
OUT=gpurun_out/r6h
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "from banet_amd import _capi; print('build id', _capi.lib().banet_build_id().decode())" 2>/dev/null | tail -1 | tee $OUT/h.txt
( timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_dense_backward.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short 2>&1 | tail -8 ) >> $OUT/h.txt
for rep in 1 2; do
  for m in 1 0; do
    BANET_ADJOINT_REUSE=$m PFRAMES=5 timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep "forward + backward" | sed "s/^/8 five-frame windows, BANET_ADJOINT_REUSE=$m: /" >> $OUT/h.txt
  done
done
timeout 600 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep "forward + backward" | sed "s/^/32 two-frame windows: /" >> $OUT/h.txt
cat $OUT/h.txt
