#!/bin/bash
# Round 4, second GPU call: round-4 tests (mask output, twin with forced mask), FETCH_SIZE of the cfg-3 gather (frame-parallel
# vs frames looped inside a wave; counters only on our kernels), infinity-cache read probe, the full default bench line with
# the new sweep gate.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
( timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q --timeout 600 -p no:cacheprovider -x -rfE --tb=short ) > $OUT/b_pytest.log 2>&1
tail -5 $OUT/b_pytest.log
hipcc --offload-arch=gfx950 -O3 tools/probe/stream_read.hip -o /tmp/stream_read 2>/dev/null
for mb in 64 128 192 512 8192; do timeout 120 /tmp/stream_read $mb; done > $OUT/b_mall_probe.txt 2>&1
grep -E "buffer|2048 workgroups" $OUT/b_mall_probe.txt
for bits in 0 4194304; do
  rm -rf /tmp/pmc_$bits
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "ba_gather|ba_syrk" --output-format csv -d /tmp/pmc_$bits -o p -- python $REPO/bench.py --frames 5 --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline --reserved $bits > $REPO/$OUT/b_pmc_$bits.log 2>&1)
  echo "pmc $bits exit $?"
  python tools/summarize_pmc.py /tmp/pmc_$bits > $OUT/b_pmc_fetch_cfg3_$bits.txt 2>&1
  grep -i "gather\|syrk" $OUT/b_pmc_fetch_cfg3_$bits.txt | head -8
done
( time timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 3 ) > $OUT/b_bench.log 2> $OUT/b_bench.err; echo "bench exit $?" >> $OUT/b_bench.err
tail -3 $OUT/b_bench.err
cp bench_detail.json $OUT/b_bench_detail.json 2>/dev/null
tail -1 $OUT/b_bench.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/b_bench_detail.json"))
for k, v in d["sweep"].items():
    p = v.get("parity", {})
    print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], "parity", p.get("ok"), p.get("max_rel_err"), "flips", p.get("mask_bits_differing"))
    for lvl, r in p.get("per_level", {}).items():
        if r.get("mask_bits_differing") or r.get("failed"):
            print("   ", lvl, json.dumps(r))
PY
exit 0
