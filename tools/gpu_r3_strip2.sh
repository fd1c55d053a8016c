#!/bin/bash
# strip gather: parity tests, phase timeline (timing builds), same-box A/B of the headline
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -x -k "strip" ) > $OUT/pytest_strip.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_strip.log
tail -15 $OUT/pytest_strip.log
rm -f $OUT/time_strip.log
for m in 2 1; do
  echo "== BANET_TIMING=$m" | tee -a $OUT/time_strip.log
  BANET_HIP_LIB=$PWD/banet_amd/lib_timing$m/libbanet_hip.so PMODE=$m timeout 300 python tools/time_strip.py 2>&1 | grep -v amdgpu | tee -a $OUT/time_strip.log
done
for tag in strip patch strip2 patch2; do
  R=0; case $tag in patch*) R=524288;; esac
  ( timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-parity --reserved $R ) > $OUT/ab_$tag.log 2> $OUT/ab_$tag.err
  python - <<PY
import json
l=[x for x in open("$OUT/ab_$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[0]); r=d["roofline"]
    print("$tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items()})
else:
    print("$tag FAILED"); print(open("$OUT/ab_$tag.err").read()[-800:])
PY
done
exit 0
