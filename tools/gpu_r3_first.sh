#!/bin/bash
# round 3, first GPU call: the new round-3 tests, the full bench line (parity on two scenes, CPU protocol, the whole sweep),
# and bench.py's own 2-rank launch on this one-GPU box (gloo override).
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=8 -s ) > $OUT/pytest_r3.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_r3.log
tail -15 $OUT/pytest_r3.log
( time timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 3 ) > $OUT/bench_full.log 2> $OUT/bench_full.err; echo "bench exit $?" >> $OUT/bench_full.err
tail -c 400 $OUT/bench_full.log; tail -5 $OUT/bench_full.err
( time BANET_BENCH_DEVICE=0 BANET_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --windows 8 ) > $OUT/bench_2rank.log 2> $OUT/bench_2rank.err; echo "2rank exit $?" >> $OUT/bench_2rank.err
tail -c 600 $OUT/bench_2rank.log; tail -5 $OUT/bench_2rank.err
exit 0
