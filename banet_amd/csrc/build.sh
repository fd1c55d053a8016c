#!/bin/bash
# Build libbanet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${BANET_BUILD_OUT:-../lib}     # BANET_BUILD_OUT: a second build (e.g. -DBANET_TIMING) next to the product library
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm"
SRCS="gather gather128 gather128p gather128s gather128q syrk syrk_wide assemble eqcon eqcon_syrk eqcon_grad solve prep sstats adjoint smallstep api"
# build id: digest of every kernel source + the compile flags (banet_build_id(); bench.py ties PMC traffic files to it).
# Only api.o depends on it, and the header is rewritten only when the digest changes.
BID=$( { cat $(ls *.hip *.hpp | sort) ../../include/banet_hip.h; echo "$FLAGS ${EXTRA_HIPCC_FLAGS:-}"; } | sha256sum | cut -c1-16)
if ! grep -qs "\"$BID\"" "$OUT/build_id.h"; then echo "#define BANET_BUILD_ID \"$BID\"" > "$OUT/build_id.h"; fi
pids=()
names=()
for f in $SRCS; do
  stale=0
  [ -f "$OUT/$f.o" ] || stale=1
  extra_dep=""
  [ "$f" = api ] && extra_dep="$OUT/build_id.h"
  for dep in "$f.hip" *.hpp ../../include/banet_hip.h $extra_dep; do
    [ "$stale" = 1 ] || { [ "$dep" -nt "$OUT/$f.o" ] && stale=1; } || true
  done
  if [ "$stale" = 1 ]; then
    echo "hipcc $f.hip"
    rm -f "$OUT/$f.o"      # a failed compile must not leave an older object for the link step
    hipcc $FLAGS ${EXTRA_HIPCC_FLAGS:-} -I"$OUT" -c "$f.hip" -o "$OUT/$f.o" &
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
# gather128s.hip keeps data in flight in v224..v255 behind the compiler's back (amdgpu_num_vgpr + inline asm): when it was rebuilt,
# check that the compiler did not allocate those registers and did not spill in the default variants (tools/check_strip_regs.py)
for n in "${names[@]:-}"; do
  if [ "$n" = gather128s ] && [ -z "${BANET_SKIP_STRIP_CHECK:-}" ] && [ -z "${EXTRA_HIPCC_FLAGS:-}" ]; then
    python3 ../../tools/check_strip_regs.py || { echo "build.sh: gather128s.hip violates its register contract" >&2; rm -f "$OUT/gather128s.o"; exit 1; }
  fi
done
OBJS=""
for f in $SRCS; do OBJS="$OBJS $OUT/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbanet_hip.so" $OBJS -ldl   # dlopen of the roctx marker library (assemble.hip)
# what this run of the script did (bench.py puts it into its line as `build_mode`, next to the driver's build_exercised record)
nsrc=$(echo $SRCS | wc -w)
echo "build_id=$BID recompiled=${#names[@]} of $nsrc objects ($(date -u +%Y-%m-%dT%H:%M:%SZ))" > "$OUT/build_mode.txt"
echo "built $OUT/libbanet_hip.so"
