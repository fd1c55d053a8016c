#!/bin/bash
# round 6: sparse training path -- tests, timing at the reference's training shape, launch list of one fused iteration
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -m gpu -q --timeout 900 -p no:cacheprovider -rfE --tb=short -k "training or sparse or fused" ) 2>&1 | tail -12 | tee $OUT/r6e_tests.txt
PB=4 PN=4096 PH=384 PW=512 timeout 600 python tools/train_graph_bench.py 2>&1 | grep -v amdgpu | tail -4 | tee $OUT/r6e_sparse_timing.txt
cd /tmp; rm -rf /tmp/prof_e
PGRAPHS=fused PB=4 PN=4096 PH=384 PW=512 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o p -- python $GRAFT_REPO_ROOT/tools/train_graph_bench.py > /tmp/prof_e.log 2>&1
f=$(find /tmp/prof_e -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/$OUT/r6e_sparse_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["Calls"]) for r in rows); ours = sum(int(r["Calls"]) for r in rows if "banet" in r["Name"])
t_all = sum(int(r["TotalDurationNs"]) for r in rows); t_ours = sum(int(r["TotalDurationNs"]) for r in rows if "banet" in r["Name"])
print("launches in the run (4 iterations + set-up): %d, of which banet:: %d; kernel time %.2f ms, banet:: %.2f ms" % (tot, ours, t_all / 1e6, t_ours / 1e6))
for r in rows[:25]:
    print("%6s x %9.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
exit 0
