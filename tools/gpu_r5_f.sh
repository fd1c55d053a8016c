#!/bin/bash
# Round 5, fifth GPU call: (1) the fused sparse-point training iteration (dense_train._SparseIteration + adj_pixel_kernel<.., SP>)
# against the lean graph and against finite differences of the float64 oracle; the dense backward tests (the kernels were touched);
# (2) the literal op at 272 < P <= 304 on the SYRK engine: parity + timing against the LDS-tiled kernel it replaces;
# (3) one training iteration, fused vs lean vs reference graph, at the reference's training shape and at the earlier bench shapes.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -q --timeout 800 -p no:cacheprovider -rfE --tb=short -k "fused or training_graph or equation or eqcon" ) > $OUT/r5f_pytest_a.log 2>&1
tail -n 25 $OUT/r5f_pytest_a.log | cut -c1-400
( timeout 900 python -m pytest tests/test_gpu_dense_backward.py -m gpu -q --timeout 800 -p no:cacheprovider -x --tb=short ) > $OUT/r5f_pytest_b.log 2>&1
tail -n 3 $OUT/r5f_pytest_b.log | cut -c1-300
for mode in 0 1; do
  echo "== BANET_EQ_LDS_KERNEL=$mode" | tee -a $OUT/r5f_eqcon.txt
  BANET_EQ_LDS_KERNEL=$mode EQ_SHAPES=8x76800x298,2x76800x298,8x4096x298,8x76800x262 timeout 600 python tools/bench_eqcon.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5f_eqcon.txt
done
for cfg in "PB=4 PN=4096 PH=384 PW=512" "PB=8 PN=4096 PH=384 PW=512" "PB=4 PH=120 PW=160" "PB=2 PH=240 PW=320"; do
  echo "== $cfg" | tee -a $OUT/r5f_train_graph.txt
  env $cfg timeout 600 python tools/train_graph_bench.py 2>&1 | grep -v amdgpu | tee -a $OUT/r5f_train_graph.txt
done
exit 0
