#!/bin/bash
# Round 6, after the evidence run: a soak of the final build on another box -- the GPU suite a second time (flakiness), the fuzzers
# with seeds the suite and the committed fuzz record do not use, and bench.py's headline three times (box / run spread).
set -u
OUT=gpurun_out/r6soak
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "from banet_amd import _capi; print('build id', _capi.lib().banet_build_id().decode())" 2>/dev/null | tail -1 | tee $OUT/soak.txt
( timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short 2>&1 | tail -4 ) >> $OUT/soak.txt
( timeout 900 python tools/fuzz_backward.py 7000 120 2>&1 | grep -v amdgpu | tail -2 ) >> $OUT/soak.txt
( timeout 900 python tools/fuzz_parity.py 9000 80 2>&1 | grep -v amdgpu | tail -3 ) >> $OUT/soak.txt
for i in 1 2 3; do
  timeout 600 python bench.py --no-sweep --no-parity --no-cpu-baseline --no-backward 2>/dev/null | grep '^{"metric"' | tail -1 | \
    python -c "import sys,json; l=json.loads(sys.stdin.read()); print('bench run $i: value %.1f  exact_syrk %.1f  frac %.4f  ms %.2f' % (l['value'], l.get('value_exact_syrk') or 0, l['roofline']['frac'], l['ms_per_step']))" >> $OUT/soak.txt
done
cat $OUT/soak.txt
