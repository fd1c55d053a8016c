#!/bin/bash
# coarse levels (<= 160x120) of a 32-window batch as two half batches on two HIP streams (BANET_SPLIT_COARSE=1) vs one batch, same box
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in 0 1 0 1; do
  BANET_SPLIT_COARSE=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-sweep --no-parity --no-cpu-baseline > $OUT/r5n_$v.log 2>&1
  python - $OUT/r5n_$v.log $v <<'PY' | tee -a $OUT/r5n_split_coarse.txt
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
d = json.loads(l[-1]); dd = json.load(open("bench_detail.json"))
print("BANET_SPLIT_COARSE=%s: %.1f LM it/s, %.3f ms per step, levels (ms, last step) %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: v.get("level_ms_last_step") for k, v in dd["roofline"]["per_level"].items()}))
PY
done
exit 0
