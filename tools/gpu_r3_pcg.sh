#!/bin/bash
# conjugate-gradient solve (solve.hip::pcg_schur_solve): the whole GPU suite, then same-box A/B vs the LDL^T (reserved_ bit 23)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x ) > $OUT/pytest_pcg.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_pcg.log
tail -12 $OUT/pytest_pcg.log
for wn in 32 8 1; do
for tag in cg ldlt cg2 ldlt2; do
  R=0; case $tag in ldlt*) R=8388608;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-parity --reserved $R --windows $wn ) > $OUT/pcg_${tag}_$wn.log 2> $OUT/pcg_${tag}_$wn.err
  python - <<PY
import json
l=[x for x in open("$OUT/pcg_${tag}_$wn.log") if x.startswith("{")]
if l:
    d=json.loads(l[0]); r=d["roofline"]
    print("B=$wn $tag", d["value"], d["ms_per_step"], {k:v["level_ms_last_step"] for k,v in r["per_level"].items()})
else:
    print("$tag FAILED"); print(open("$OUT/pcg_${tag}_$wn.err").read()[-800:])
PY
done
done
exit 0
