#!/bin/bash
# kernel statistics of the dense training step (32 windows) + B = 256 A/B of the strip segment heights vs the patch kernel
set -u
OUT=gpurun_out; mkdir -p $OUT $OUT/prof_train; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
rm -rf /tmp/prof_train && mkdir -p /tmp/prof_train
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o tr -- python $REPO/tools/bench_dense_train.py 32 480 640 2 > $REPO/$OUT/train_prof_run.log 2>&1)
for f in $(find /tmp/prof_train -name "*kernel_stats.csv"); do cp "$f" $OUT/prof_train/; done
tail -5 $OUT/train_prof_run.log
head -30 $OUT/prof_train/*kernel_stats.csv | cut -c1-200
for tag in strip seg16 patch; do
  R=0; case $tag in patch) R=524288;; seg16) R=2097152;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-sweep --no-cpu-baseline --no-parity --reserved $R --windows 256 ) > $OUT/ab256_$tag.log 2> $OUT/ab256_$tag.err
  python - <<PY
import json
l=[x for x in open("$OUT/ab256_$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[0]); r=d["roofline"]
    print("B=256 $tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items()})
else:
    print("$tag FAILED"); print(open("$OUT/ab256_$tag.err").read()[-800:])
PY
done
exit 0
