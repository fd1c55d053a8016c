#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x -k "strip" ) 2>&1 | tail -3
for tag in seg32 seg16 patch seg32b seg16b patchb; do
  R=0; case $tag in seg16*) R=2097152;; patch*) R=524288;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-parity --reserved $R ) > $OUT/seg_$tag.log 2> $OUT/seg_$tag.err
  python - <<PY
import json
l=[x for x in open("$OUT/seg_$tag.log") if x.startswith("{")]
d=json.loads(l[0]); r=d["roofline"]
print("B=32 $tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items()})
PY
done
exit 0
