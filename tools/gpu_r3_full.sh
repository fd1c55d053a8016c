#!/bin/bash
# full GPU test suite + smoke + the full bench line
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=5 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 3 ) > $OUT/bench_full.log 2> $OUT/bench_full.err; echo "bench exit $?" >> $OUT/bench_full.err
tail -4 $OUT/bench_full.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench_full.log") if x.startswith("{")]
d=json.loads(l[0])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"])
print("parity", d["parity"]["ok"], d["parity"]["max_rel_err"], [(s["window"], s["max_rel_err"]) for s in d["parity"]["scenes"]])
for k,v in d["sweep"].items():
    print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"], v["parity"]["ok"], v["parity"]["max_rel_err"])
    if not v["parity"]["ok"]:
        print(json.dumps({lv:{kk:vv for kk,vv in r.items() if "step" in kk or "mask" in kk or kk=="failed"} for lv,r in v["parity"]["per_level"].items()}))
PY
exit 0
