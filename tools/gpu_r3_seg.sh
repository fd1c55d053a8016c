#!/bin/bash
# strip gather: 32-row vs 16-row segments at the headline batch (same box, alternating), cfg-3 likewise
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for tag in seg32 seg16 seg32b seg16b; do
  R=0; case $tag in seg16*) R=2097152;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-parity --reserved $R ) > $OUT/seg_$tag.log 2> $OUT/seg_$tag.err
  python - <<PY
import json
l=[x for x in open("$OUT/seg_$tag.log") if x.startswith("{")]
d=json.loads(l[0]); r=d["roofline"]
print("B=32 $tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items()})
PY
done
for tag in seg32 seg16; do
  R=0; case $tag in seg16*) R=2097152;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --no-parity --frames 5 --reserved $R ) > $OUT/seg5_$tag.log 2> $OUT/seg5_$tag.err
  python - <<PY
import json
l=[x for x in open("$OUT/seg5_$tag.log") if x.startswith("{")]
d=json.loads(l[0]); r=d["roofline"]
print("cfg3 $tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items()})
PY
done
exit 0
