#!/bin/bash
# PMC traffic passes for a new build id (the adjoint plan changed; the forward kernels are untouched): headline, cfg-3, cfg-5 share ->
# pmc_traffic.json; the backward / sparse tests; one short bench line that must accept the traffic file.
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
pmc_pass() {   # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$c
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "ba_gather|ba_syrk" --output-format csv -d /tmp/pmc_${name}_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline "$@" > $REPO/$OUT/pmc_${name}_$c.log 2>&1)
    echo "$name $c exit $?"
  done
  { python tools/summarize_pmc.py /tmp/pmc_${name}_FETCH_SIZE; python tools/summarize_pmc.py /tmp/pmc_${name}_WRITE_SIZE; } > $OUT/pmc_fetch_write_$name.txt 2>&1
}
pmc_pass headline
python tools/make_pmc_traffic.py /tmp/pmc_headline_FETCH_SIZE /tmp/pmc_headline_WRITE_SIZE 32 profiles/pmc_traffic.json > /dev/null
pmc_pass cfg3 --frames 5
python tools/make_pmc_traffic.py /tmp/pmc_cfg3_FETCH_SIZE /tmp/pmc_cfg3_WRITE_SIZE 32 profiles/pmc_traffic.json cfg3_5frame_B32 5 480 640 128 10 > /dev/null
pmc_pass cfg5 --frames 8 --height 960 --width 1280 --basis 256 --iters 15 --windows 8
python tools/make_pmc_traffic.py /tmp/pmc_cfg5_FETCH_SIZE /tmp/pmc_cfg5_WRITE_SIZE 8 profiles/pmc_traffic.json cfg5_8frame_1280x960_K256_B8 8 960 1280 256 15 > /dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python - <<'PY'
import json
d = json.load(open("profiles/pmc_traffic.json"))
print("build", d["build_id"], "traffic headline", d["hbm_bytes_per_launch"], d["overfetch"])
for k, v in d.get("workloads", {}).items():
    print("traffic", k, v["build_id"], v["hbm_bytes_per_launch"], v["overfetch"])
PY
( timeout 600 python -m pytest tests/test_gpu_dense_backward.py tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -q --timeout 500 -p no:cacheprovider --tb=short -k "backward or fused or training_graph or adjoint or differentiable" ) > $OUT/r5p_pytest.log 2>&1
tail -n 3 $OUT/r5p_pytest.log | cut -c1-200
timeout 600 python bench.py --steps 5 --warmup 2 --no-sweep --no-parity --no-cpu-baseline > $OUT/r5p_bench.log 2>&1
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r5p_bench.log") if x.startswith("{")]
d = json.loads(l[-1]); print("bench", d["value"], d["build_id"], "traffic", d["roofline"]["traffic"])
PY
exit 0
