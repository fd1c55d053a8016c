#!/bin/bash
# batch 1 / 2: strip gather in a single resident round vs the tile kernels (reserved_ bit 22 switches it off)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for wn in 1 2; do
for tag in strip tiles strip2 tiles2; do
  R=0; case $tag in tiles*) R=4194304;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-sweep --no-cpu-baseline --no-parity --reserved $R --windows $wn ) > $OUT/b1_${tag}_$wn.log 2> $OUT/b1_${tag}_$wn.err
  python - <<PY
import json
l=[x for x in open("$OUT/b1_${tag}_$wn.log") if x.startswith("{")]
if l:
    d=json.loads(l[0]); r=d["roofline"]
    print("B=$wn $tag", d["value"], d["ms_per_step"], {k:(v["gather_avg_us"], v["syrk_avg_us"], v["level_ms_last_step"]) for k,v in r["per_level"].items()})
else:
    print("$tag FAILED"); print(open("$OUT/b1_${tag}_$wn.err").read()[-800:])
PY
done
done
exit 0
