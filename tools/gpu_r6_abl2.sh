#!/bin/bash
# Round 6, late: what bounds adj_pixel2 / adj_tile2 after the instruction-count pass -- same box, the product library against four
# ablation builds (-DBANET_ADJ_ABLATE=1: pixel kernel's texels from 12 fixed rows; 2: its dsrc / dbasis stores never execute;
# 4: tile kernel's texel / source rows from a fixed neighbourhood; 8: one LDS read-modify-write per visit instead of twelve)
set -u
OUT=gpurun_out/r6abl2
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
run() {  # label, lib
  rm -rf /tmp/prof_t
  (cd /tmp && BANET_HIP_LIB=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $GRAFT_REPO_ROOT/tools/bench_dense_train.py 32 480 640 2 > /tmp/prof_t.log 2>&1)
  grep "forward + backward" /tmp/prof_t.log | sed "s/^/$1: /"
  for f in $(find /tmp/prof_t -name "*kernel_stats.csv"); do python - "$f" "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'adj_pixel2' in r['Name'] or 'adj_tile2' in r['Name'] or 'adj_basis6' in r['Name']:
        print("%s: %-60s calls=%4s avg_us=%9.1f total_ms=%8.2f" % (sys.argv[2], r['Name'][33:93], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  done
}
{ run product $PWD/banet_amd/lib/libbanet_hip.so
  for n in 1 2 4 8; do run ablate$n $PWD/banet_amd/lib_abl$n/libbanet_hip.so; done
  run product_again $PWD/banet_amd/lib/libbanet_hip.so; } > $OUT/abl.txt 2>&1
cat $OUT/abl.txt
