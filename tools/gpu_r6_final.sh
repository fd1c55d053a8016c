#!/bin/bash
# Round 6, final artefacts in one gpurun call (everything on the library this snapshot carries; its build id goes into the file names):
#   GPU tests + smoke; PMC traffic passes of the headline / cfg-3 / cfg-5 share -> profiles/pmc_traffic.json; rocprofv3 kernel stats of
#   the bench command; bench.py with default arguments (the driver's shape) and its 2-rank launch on this one-GPU box (gloo; carries
#   the cfg-4 / cfg-5 dpN entries); the dense training step (timing, kernel stats, FETCH / WRITE counters of the backward kernels);
#   the sparse training iteration and the reference's own sparse tracker; the literal op at P = 262 / 298.
set -u
OUT=gpurun_out/r6f
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
REPO=$PWD
BID=$(python -c "from banet_amd import _capi; print(_capi.lib().banet_build_id().decode())" 2>/dev/null | tail -1)
echo "build id $BID" | tee $OUT/build_id.txt
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -rfE --tb=short --durations=5 ) > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -12 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?" >> $OUT/smoke.txt
tail -2 $OUT/smoke.txt
pmc_pass() {   # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$c
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "ba_gather|ba_syrk" --output-format csv -d /tmp/pmc_${name}_$c -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-sweep --no-parity --no-cpu-baseline --no-exact-syrk --no-backward "$@" > $REPO/$OUT/pmc_${name}_$c.log 2>&1)
    echo "$name $c exit $?"
  done
  { python tools/summarize_pmc.py /tmp/pmc_${name}_FETCH_SIZE; python tools/summarize_pmc.py /tmp/pmc_${name}_WRITE_SIZE; } > $OUT/pmc_fetch_write_$name.txt 2>&1
}
pmc_pass headline
python tools/make_pmc_traffic.py /tmp/pmc_headline_FETCH_SIZE /tmp/pmc_headline_WRITE_SIZE 32 profiles/pmc_traffic.json > /dev/null
pmc_pass cfg3 --frames 5
python tools/make_pmc_traffic.py /tmp/pmc_cfg3_FETCH_SIZE /tmp/pmc_cfg3_WRITE_SIZE 32 profiles/pmc_traffic.json cfg3_5frame_B32 5 480 640 128 10 > /dev/null
pmc_pass cfg5 --frames 8 --height 960 --width 1280 --basis 256 --iters 15 --windows 8
python tools/make_pmc_traffic.py /tmp/pmc_cfg5_FETCH_SIZE /tmp/pmc_cfg5_WRITE_SIZE 8 profiles/pmc_traffic.json cfg5_8frame_1280x960_K256_B8 8 960 1280 256 15 > /dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o fin -- python $REPO/bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-sweep --no-exact-syrk --no-backward > $REPO/$OUT/prof_run.log 2>&1)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" $OUT/bench_kernel_stats_build_$BID.csv; done
( time timeout 1800 python bench.py ) > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/bench_default.err
tail -4 $OUT/bench_default.err
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_default_line.json
( BANET_BENCH_DEVICE=0 BANET_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 ) > $OUT/bench_2rank.log 2> $OUT/bench_2rank.err; echo "2rank exit $?" >> $OUT/bench_2rank.err
grep '^{"metric"' $OUT/bench_2rank.log | tail -1 > $OUT/bench_2rank_gloo_one_gpu_line.json
tail -1 $OUT/bench_2rank.err
# ---- the dense training step
{ for w in 32 8 2; do timeout 600 python tools/bench_dense_train.py $w 480 640 2 2>&1 | grep -v amdgpu | tail -3; done
  PFRAMES=5 timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3
  echo "--- 8 five-frame windows with BANET_ADJOINT_REUSE=0 (every target frame's call recomputes z2 / zeta / e)"
  BANET_ADJOINT_REUSE=0 PFRAMES=5 timeout 600 python tools/bench_dense_train.py 8 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2
  echo "--- A/B on this box: BANET_ADJOINT_FOLD=0 (round-5 rows + per-texel gather), BANET_SMALL_STEP_HIP=0 (torch small step)"
  BANET_ADJOINT_FOLD=0 timeout 600 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2
  BANET_SMALL_STEP_HIP=0 timeout 600 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2
  BANET_ADJOINT_TILE=1 timeout 600 python tools/bench_dense_train.py 32 480 640 2 2>&1 | grep -v amdgpu | tail -3 | head -2 | sed 's/^/(BANET_ADJOINT_TILE=1: adj_tile_kernel, one visit per wave instruction) /'
} > $OUT/dense_train.txt 2>&1
cat $OUT/dense_train.txt | head -20
rm -rf /tmp/prof_t
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $REPO/tools/bench_dense_train.py 32 480 640 2 > /tmp/prof_t.log 2>&1)
for f in $(find /tmp/prof_t -name "*kernel_stats.csv"); do cp "$f" $OUT/dense_train_kernel_stats_build_$BID.csv; done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bp_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "adj_|small_|spd_" --output-format csv -d /tmp/bp_$c -o p -- python $REPO/tools/bench_dense_train.py 32 480 640 2 > /tmp/bp_$c.log 2>&1)
done
{ python tools/summarize_pmc.py /tmp/bp_FETCH_SIZE; python tools/summarize_pmc.py /tmp/bp_WRITE_SIZE; } > $OUT/backward_pmc_fetch_write.txt 2>&1
# ---- the sparse training iteration (the reference's training shape) and the reference's own tracker workload
{ PB=4 PN=4096 PH=384 PW=512 timeout 600 python tools/train_graph_bench.py 2>&1 | grep -v amdgpu | tail -3
  PB=8 PN=4096 PH=384 PW=512 timeout 600 python tools/train_graph_bench.py 2>&1 | grep -v amdgpu | tail -3
  timeout 600 python tools/bench_sparse.py 2>&1 | grep -v amdgpu | tail -2; } > $OUT/sparse_training_and_tracker.txt 2>&1
rm -rf /tmp/prof_s
(cd /tmp && PGRAPHS=fused PB=4 PN=4096 PH=384 PW=512 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o p -- python $REPO/tools/train_graph_bench.py > /tmp/prof_s.log 2>&1)
for f in $(find /tmp/prof_s -name "*kernel_stats.csv"); do cp "$f" $OUT/sparse_training_iteration_kernel_stats_build_$BID.csv; done
# ---- the literal op at K = 256 shapes
EQ_SHAPES=8x76800x262,8x76800x298,8x76800x134 timeout 600 python tools/bench_eqcon.py 2>&1 | grep "^B=" > $OUT/eqcon_literal_op.txt
cat $OUT/eqcon_literal_op.txt
exit 0
