#!/bin/bash
# Build libbanet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for f in gather gather128 gather128p syrk syrk_wide assemble eqcon eqcon_syrk eqcon_grad solve prep sstats api; do
  if [ ! -f "$OUT/$f.o" ] || [ "$f.hip" -nt "$OUT/$f.o" ] || [ common.hpp -nt "$OUT/$f.o" ] || [ kernels.hpp -nt "$OUT/$f.o" ] || [ gather_common.hpp -nt "$OUT/$f.o" ] || [ syrk_split.hpp -nt "$OUT/$f.o" ] || [ ../../include/banet_hip.h -nt "$OUT/$f.o" ]; then
    echo "hipcc $f.hip"
    hipcc $FLAGS ${EXTRA_HIPCC_FLAGS:-} -c "$f.hip" -o "$OUT/$f.o" &
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbanet_hip.so" "$OUT/gather.o" "$OUT/gather128.o" "$OUT/gather128p.o" "$OUT/syrk.o" "$OUT/syrk_wide.o" "$OUT/assemble.o" "$OUT/eqcon.o" "$OUT/eqcon_syrk.o" "$OUT/eqcon_grad.o" "$OUT/solve.o" "$OUT/prep.o" "$OUT/sstats.o" "$OUT/api.o"
echo "built $OUT/libbanet_hip.so"
