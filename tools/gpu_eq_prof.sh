#!/bin/bash
# per-kernel times of the literal op (EquationConstruction / Grad) at one wide and one narrow shape
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
for sh in 8x76800x262 8x76800x134; do
  rm -rf /tmp/eqp_$sh
  EQ_SHAPES=$sh timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eqp_$sh -o eq -- python $GRAFT_REPO_ROOT/tools/bench_eqcon.py 2>&1 | grep "^B=" | tee -a $GRAFT_REPO_ROOT/$OUT/eq_prof.log
  f=$(find /tmp/eqp_$sh -name "*kernel_stats.csv" | head -1)
  echo "== $sh" >> $GRAFT_REPO_ROOT/$OUT/eq_prof.log
  grep -E "banet|Name" $f | cut -d, -f1-4,6,7 | cut -c1-200 >> $GRAFT_REPO_ROOT/$OUT/eq_prof.log
done
cat $GRAFT_REPO_ROOT/$OUT/eq_prof.log
