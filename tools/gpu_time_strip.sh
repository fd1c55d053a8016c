#!/bin/bash
# phase timeline of the strip gather from the -DBANET_TIMING=1 / =2 builds (banet_amd/lib_timing1, lib_timing2) + PMC traffic
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
for m in 2 1; do
  echo "== BANET_TIMING=$m" | tee -a $OUT/time_strip.log
  BANET_HIP_LIB=$PWD/banet_amd/lib_timing$m/libbanet_hip.so PMODE=$m timeout 300 python tools/time_strip.py 2>&1 | grep -v amdgpu | tee -a $OUT/time_strip.log
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline > $REPO/$OUT/pmc_$c.log 2>&1)
  echo "$c exit $?"
done
python tools/make_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 32 $OUT/pmc_traffic.json
{ python tools/summarize_pmc.py /tmp/pmc_FETCH_SIZE; python tools/summarize_pmc.py /tmp/pmc_WRITE_SIZE; } > $OUT/pmc_fetch_write.txt 2>&1
cat $OUT/pmc_fetch_write.txt | head -40
exit 0
