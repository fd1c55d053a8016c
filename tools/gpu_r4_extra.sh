set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD; OUT=gpurun_out; mkdir -p $OUT/prof_cfg3
rm -rf /tmp/prof3 && mkdir -p /tmp/prof3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o c3 -- python $REPO/bench.py --frames 5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sweep > $REPO/$OUT/x_prof_cfg3.log 2>&1)
for f in $(find /tmp/prof3 -name "*kernel_stats.csv"); do cp "$f" $OUT/prof_cfg3/; done
grep "banet" $OUT/prof_cfg3/*kernel_stats.csv | cut -c1-170 | head -12
timeout 600 python bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --no-parity > $OUT/x_bench.log 2>&1; tail -1 $OUT/x_bench.log | cut -c1-400
