#!/bin/bash
# Round 4 (experiment for round 5): the strip gather leaves the fp16 SYRK's record maxima in its partial rows (no ba_recmax_kernel
# pass): side build in banet_amd/lib_gm (BANET_HIP_LIB) against the in-tree library, same box.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
GM=$PWD/banet_amd/lib_gm/libbanet_hip.so
( BANET_HIP_LIB=$GM timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short -k "f16 or cfg3 or cfg5 or strip or five_level" ) > $OUT/gm_pytest.log 2>&1
tail -4 $OUT/gm_pytest.log
show() {
python - "$@" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "NO JSON LINE"); sys.exit(0)
d = json.loads(l[-1]); r = d["roofline"]
print(sys.argv[2], d["build_id"], d["value"], d["ms_per_step"], "syrk", r["syrk_kernel"]["avg_launch_us"], "parity", (d.get("parity") or {}).get("max_rel_err"))
PY
}
for rep in 1 2; do
  timeout 600 python bench.py --steps 8 --warmup 3 --no-sweep --no-cpu-baseline > $OUT/gm_base_$rep.log 2>&1; show $OUT/gm_base_$rep.log "headline in-tree"
  BANET_HIP_LIB=$GM timeout 600 python bench.py --steps 8 --warmup 3 --no-sweep --no-cpu-baseline > $OUT/gm_new_$rep.log 2>&1; show $OUT/gm_new_$rep.log "headline gather-maxima"
done
timeout 600 python bench.py --frames 5 --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --no-parity > $OUT/gm_c3_base.log 2>&1; show $OUT/gm_c3_base.log "cfg3 in-tree"
BANET_HIP_LIB=$GM timeout 600 python bench.py --frames 5 --steps 3 --warmup 1 --no-sweep --no-cpu-baseline --no-parity > $OUT/gm_c3_new.log 2>&1; show $OUT/gm_c3_new.log "cfg3 gather-maxima"
exit 0
