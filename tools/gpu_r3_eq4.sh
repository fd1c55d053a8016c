#!/bin/bash
# literal op, 144 < P <= 272: the four jobs side by side in one launch vs one launch each (BANET_EQ_FOUR_PASS=1)
OUT=gpurun_out/r3_eq4; mkdir -p $OUT
timeout 600 python -m pytest tests/ -x -q -m gpu -k "equation or eqcon or literal" 2>&1 | tail -5 | tee $OUT/tests.log
export EQ_SHAPES=2x76800x262,8x76800x262,8x76800x200,8x76800x150,8x76800x134
echo "== one launch (default)" | tee $OUT/bench.log
timeout 300 python tools/bench_eqcon.py 2>&1 | grep "^B=" | tee -a $OUT/bench.log
echo "== four launches (BANET_EQ_FOUR_PASS=1)" | tee -a $OUT/bench.log
BANET_EQ_FOUR_PASS=1 timeout 300 python tools/bench_eqcon.py 2>&1 | grep "^B=" | tee -a $OUT/bench.log
