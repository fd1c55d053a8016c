#!/bin/bash
# L2 -> fabric read requests of the gather kernel for two reserved_ bit settings (A/B of a locality experiment).
# usage: bash tools/pmc_tcc_ab.sh "0" "2048"
export PB=${PB:-8} PROUNDS=1
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
for bits in "$@"; do
  rm -rf /tmp/pmc_ab
  (cd /tmp && PBITS=$bits timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_ab -o p -- python $REPO/tools/prof_assemble.py > $REPO/$OUT/pmc_ab_$bits.log 2>&1)
  echo "== bits $bits"
  python tools/summarize_pmc.py /tmp/pmc_ab 2>/dev/null | grep -E "gather" | cut -c1-160
done | tee $OUT/pmc_tcc_ab.txt
