#!/bin/bash
# Round 6, late: instruction-count pass over adj_pixel2 / adj_tile2 (32-bit byte offsets from per-window scalar bases, the symmetric
# seed block, the 2 x 6 Jacobian algebra on row pairs): backward tests + the training step, A/B against the committed numbers.
set -u
OUT=gpurun_out/r6g
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "from banet_amd import _capi; print('build id', _capi.lib().banet_build_id().decode())" 2>/dev/null | tail -1 | tee $OUT/g.txt
( timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_dense_backward.py tests/test_gpu_round5.py -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short 2>&1 | tail -15 ) >> $OUT/g.txt
for w in 32 8; do timeout 600 python tools/bench_dense_train.py $w 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2 >> $OUT/g.txt; done
PFRAMES=5 timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2 >> $OUT/g.txt
rm -rf /tmp/prof_t
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $GRAFT_REPO_ROOT/tools/bench_dense_train.py 32 480 640 2 > /tmp/prof_t.log 2>&1)
for f in $(find /tmp/prof_t -name "*kernel_stats.csv"); do python - "$f" >> $OUT/g.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'adj_' in r['Name'] or 'small_' in r['Name'] or 'gather' in r['Name']:
        print("%-86s calls=%4s avg_us=%9.1f total_ms=%8.2f" % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
timeout 900 python tools/fuzz_backward.py 8100 40 2>&1 | grep -v amdgpu | tail -1 >> $OUT/g.txt
cat $OUT/g.txt
