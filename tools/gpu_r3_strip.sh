#!/bin/bash
# round 3: the strip gather -- parity tests, then same-box A/B of the headline (strip on the 640x480 level vs the patch kernel)
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x -k "strip" ) > $OUT/pytest_strip.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_strip.log
tail -30 $OUT/pytest_strip.log
for tag in strip patch strip2 patch2; do
  R=0; case $tag in patch*) R=524288;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-parity --reserved $R ) > $OUT/ab_$tag.log 2> $OUT/ab_$tag.err
  echo "$tag exit $?"
  python - <<PY
import json
l=[x for x in open("$OUT/ab_$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[0]); r=d["roofline"]
    print("$tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items()})
PY
done
( timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline ) > $OUT/bench_strip_parity.log 2> $OUT/bench_strip_parity.err; echo "parity bench exit $?"
tail -c 1500 $OUT/bench_strip_parity.err
exit 0
