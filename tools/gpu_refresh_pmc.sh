#!/bin/bash
# after a change that does not touch the forward kernels: GPU tests + the PMC traffic passes for the new build id (a short form of
# gpu_final_round.sh; the bench line itself is unchanged)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3 ) | tee $OUT/pytest_gpu_short.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline > $REPO/$OUT/pmc_$c.log 2>&1)
  echo "$c exit $?"
done
python tools/make_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 32 $OUT/pmc_traffic.json | cut -c1-400
timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'parity', d['parity']['ok'], d['parity']['max_rel_err'])
"
