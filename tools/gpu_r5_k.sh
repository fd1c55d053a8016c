#!/bin/bash
# Round 5, last call: random-shape fuzz of the forward and of the fused dense backward on the final build, the fused sparse training
# iteration at three more shapes.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 600 python tools/fuzz_parity.py 31000 60 2>&1 | grep -v amdgpu | tail -n 3; echo "fuzz exit $?" ) | tee $OUT/r5k_fuzz.txt
( timeout 600 python tools/fuzz_backward.py 700 40 2>&1 | grep -v amdgpu | tail -n 3; echo "fuzz_bwd exit $?" ) | tee -a $OUT/r5k_fuzz.txt
( timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q --timeout 500 -p no:cacheprovider -rfE --tb=short -k fused ) > $OUT/r5k_pytest.log 2>&1
tail -n 5 $OUT/r5k_pytest.log | cut -c1-300 | tee -a $OUT/r5k_fuzz.txt
exit 0
