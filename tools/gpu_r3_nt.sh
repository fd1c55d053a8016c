#!/bin/bash
# strip gather: nt hint on the source-feature loads (lib_nt1) / on the window rows too (lib_nt2) vs the product library
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for tag in base nt1 nt2 base_b nt1_b nt2_b; do
  L=$PWD/banet_amd/lib/libbanet_hip.so
  case $tag in nt1*) L=$PWD/banet_amd/lib_nt1/libbanet_hip.so;; nt2*) L=$PWD/banet_amd/lib_nt2/libbanet_hip.so;; esac
  ( BANET_HIP_LIB=$L timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-parity ) > $OUT/nt_$tag.log 2> $OUT/nt_$tag.err
  python - <<PY
import json
l=[x for x in open("$OUT/nt_$tag.log") if x.startswith("{")]
if l:
    d=json.loads(l[0]); r=d["roofline"]
    print("$tag", d["value"], d["ms_per_step"], r["frac"], {k:(v["gather_avg_us"], v["syrk_avg_us"]) for k,v in r["per_level"].items() if k in ("320x240","640x480")})
else:
    print("$tag FAILED"); print(open("$OUT/nt_$tag.err").read()[-600:])
PY
done
exit 0
