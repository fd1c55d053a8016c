#!/bin/bash
# Build libbanet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${BANET_BUILD_OUT:-../lib}     # BANET_BUILD_OUT: a second build (e.g. -DBANET_TIMING) next to the product library
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
SRCS="gather gather128 gather128p syrk syrk_wide assemble eqcon eqcon_syrk eqcon_grad solve prep sstats adjoint api"
pids=()
names=()
for f in $SRCS; do
  stale=0
  [ -f "$OUT/$f.o" ] || stale=1
  for dep in "$f.hip" *.hpp ../../include/banet_hip.h; do
    [ "$stale" = 1 ] || { [ "$dep" -nt "$OUT/$f.o" ] && stale=1; } || true
  done
  if [ "$stale" = 1 ]; then
    echo "hipcc $f.hip"
    rm -f "$OUT/$f.o"      # a failed compile must not leave an older object for the link step
    hipcc $FLAGS ${EXTRA_HIPCC_FLAGS:-} -c "$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
    names+=("$f")
  fi
done
fail=0
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}"; then
    echo "build.sh: hipcc failed on ${names[$i]}.hip" >&2
    fail=1
  fi
done
[ "$fail" = 0 ] || exit 1
OBJS=""
for f in $SRCS; do OBJS="$OBJS $OUT/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbanet_hip.so" $OBJS
echo "built $OUT/libbanet_hip.so"
