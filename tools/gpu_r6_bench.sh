#!/bin/bash
# round 6: bench.py as the driver runs it (default arguments) + the 2-rank line on one GPU (gloo) with the dp sweep entries
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 1500 python bench.py ) > $OUT/r6_bench_default.log 2>&1
tail -1 $OUT/r6_bench_default.log | head -c 4200 > $OUT/r6_bench_default_line.json; echo >> $OUT/r6_bench_default_line.json
grep -E "^real|Error|FAILED|Traceback" $OUT/r6_bench_default.log | head
cp bench_detail.json $OUT/r6_bench_detail.json 2>/dev/null
( time BANET_BENCH_DEVICE=0 BANET_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 3 --warmup 1 ) > $OUT/r6_bench_2rank.log 2>&1
tail -1 $OUT/r6_bench_2rank.log | head -c 4200 > $OUT/r6_bench_2rank_line.json; echo >> $OUT/r6_bench_2rank_line.json
grep -E "^real|Error|FAILED|Traceback" $OUT/r6_bench_2rank.log | head
exit 0
