#!/bin/bash
# strip gather with 16-row segments on mid-size launches: parity tests, same-box A/B at 32 and 8 windows
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x -k "strip" ) > $OUT/pytest_strip.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_strip.log
tail -8 $OUT/pytest_strip.log
for wn in 32 8; do
for tag in strip patch strip2 patch2; do
  R=0; case $tag in patch*) R=524288;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-parity --reserved $R --windows $wn ) > $OUT/ab_${tag}_$wn.log 2> $OUT/ab_${tag}_$wn.err
  python - <<PY
import json
l=[x for x in open("$OUT/ab_${tag}_$wn.log") if x.startswith("{")]
if l:
    d=json.loads(l[0]); r=d["roofline"]
    print("B=$wn $tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items()})
else:
    print("$tag FAILED"); print(open("$OUT/ab_${tag}_$wn.err").read()[-800:])
PY
done
done
exit 0
