#!/bin/bash
# Sanitizer pass over the HOST side of libbanet_hip.so ("sanitizers"): every translation unit is rebuilt with
# AddressSanitizer + UndefinedBehaviorSanitizer on the host half only (-Xarch_host; device code is unchanged), linked
# into build/asan/libbanet_hip_asan.so, and every entry point that works without a GPU -- argument validation, workspace
# queries, the plan functions behind them over a sweep of shapes, error strings, parameter defaults -- is driven from a
# plain-C program.  Needs no GPU.  Usage: bash tools/asan_capi.sh   (writes profiles/r02_asan_capi.txt)
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=build/asan
mkdir -p $OUT
SRCS="gather gather128 gather128p syrk syrk_wide assemble eqcon eqcon_syrk eqcon_grad solve prep sstats adjoint api"
SAN="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined"
pids=()
for f in $SRCS; do
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wall -Wno-unused-function $SAN -c banet_amd/csrc/$f.hip -o $OUT/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""; for f in $SRCS; do OBJS="$OBJS $OUT/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/libbanet_hip_asan.so $OBJS
cat > $OUT/drive.c <<'EOC'
#include <stdio.h>
#include <string.h>
#include "banet_hip.h"
static int checks = 0;
#define EXPECT(c) do { ++checks; if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main(void) {
  EXPECT(banet_version() == BANET_VERSION);
  for (int code = 1; code >= -6; --code) EXPECT(banet_error_string(code) != 0 && strlen(banet_error_string(code)) > 0);
  banet_lm_params_t lp; banet_lm_params_default(&lp); banet_lm_params_default(0);
  EXPECT(lp.solver == BANET_SOLVER_QR && lp.residual_ratio == 1.0f);
  /* workspace / plan queries over a sweep of shapes (the host-side planning code is where the index arithmetic lives) */
  const int Ps[] = {1, 6, 7, 38, 70, 134, 143, 144, 145, 262, 272, 273};
  const int Ns[] = {1, 63, 64, 65, 4096, 76800, 307200};
  for (unsigned i = 0; i < sizeof Ps / sizeof *Ps; ++i)
    for (unsigned j = 0; j < sizeof Ns / sizeof *Ns; ++j)
      for (int B = 1; B <= 33; B += 8) {
        size_t a = banet_equation_construction_workspace_bytes(B, Ns[j], 128, Ps[i]);
        size_t b = banet_equation_construction_grad_workspace_bytes(B, Ns[j], 128, Ps[i]);
        EXPECT((Ps[i] <= 272) == (a > 0));
        (void)b;
      }
  EXPECT(banet_equation_construction_workspace_bytes(0, 1, 1, 1) == 0);
  EXPECT(banet_equation_construction_f32(0, 0, 0, 0, 0, 1, 8, 4, 6, 0, 0, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_equation_construction_grad_f32(0, 0, 0, 0, 0, 0, 0, 0, 1, 8, 4, 6, 0, 0, 0) == BANET_ERR_INVALID_ARG);
  const int dims[][2] = {{30, 40}, {60, 80}, {120, 160}, {240, 320}, {480, 640}, {960, 1280}, {37, 53}, {8, 8}, {1, 1}};
  const int Ks[] = {0, 4, 16, 32, 64, 128, 200, 256, 257};
  float dummy[16];
  for (unsigned d = 0; d < sizeof dims / sizeof *dims; ++d)
    for (unsigned k = 0; k < sizeof Ks / sizeof *Ks; ++k)
      for (int pairs = 0; pairs <= 7; pairs += (pairs < 2 ? 1 : 3))
        for (int B = 1; B <= 257; B = B * 4 + 1)
          for (int C = 8; C <= 256; C *= 4) {
            banet_level_t lv; memset(&lv, 0, sizeof lv);
            lv.B = B; lv.H = dims[d][0]; lv.W = dims[d][1]; lv.N = lv.H * lv.W; lv.C = C == 32 ? 128 : C; lv.K = Ks[k];
            lv.variant = Ks[k] ? BANET_BUNDLE : BANET_BUNDLE_CAMERA; lv.dense = 1; lv.scale = 1.0f; lv.pairs = pairs;
            lv.normalize_rays = 1;
            size_t a = banet_ba_assemble_workspace_bytes(&lv), l = banet_lm_level_workspace_bytes(&lv);
            EXPECT((a == 0) == (l == 0)); EXPECT(l >= a); EXPECT(a % 256 == 0 && l % 256 == 0);
            if (Ks[k] == 257) EXPECT(a == 0);
            /* the entry points reject the level before any launch: null tensors */
            EXPECT(banet_ba_assemble_f32(&lv, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 1u << 20, 0) == BANET_ERR_INVALID_ARG);
            banet_state_t st; memset(&st, 0, sizeof st);
            EXPECT(banet_lm_level_ex_f32(&lv, 0, 1.0f, 3, 1, &lp, &st, dummy, 1u << 20, 0) == BANET_ERR_INVALID_ARG);
            /* backward of the dense assembly: supported-set query + plan arithmetic, rejection before any launch */
            size_t adj = banet_dense_adjoint_workspace_bytes(&lv);
            EXPECT((adj > 0) == (Ks[k] >= 1 && Ks[k] <= 128 && pairs <= 1)); EXPECT(adj % 256 == 0);
            EXPECT(banet_dense_adjoint_f32(&lv, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy,
                                           1u << 20, 0) == BANET_ERR_INVALID_ARG);   /* null level tensors */
            EXPECT(banet_sample_stats_grad_workspace_bytes(B, 4096, lv.C, lv.H, lv.W) % 256 == 0);   /* 0 when B H W 3C >= 2^32 */
          }
  /* sparse (reference-layout) levels */
  for (int N = 1; N <= 8192; N = N * 3 + 1) {
    banet_level_t lv; memset(&lv, 0, sizeof lv);
    lv.B = 2; lv.N = N; lv.C = 64; lv.K = 0; lv.H = 60; lv.W = 80; lv.variant = BANET_LEGACY_LM; lv.tgt_has_grad = 1;
    EXPECT(banet_lm_level_workspace_bytes(&lv) > 0);
  }
  banet_lm_params_t bad = lp; bad.solver = 7;
  { banet_level_t lv; memset(&lv, 0, sizeof lv); banet_state_t st; memset(&st, 0, sizeof st);
    EXPECT(banet_lm_level_ex_f32(&lv, 0, 1.0f, 1, 1, &bad, &st, 0, 0, 0) == BANET_ERR_INVALID_ARG); }
  EXPECT(banet_resample_f32(0, 0, 0, 1, 1, 1, 1, 1, BANET_RESAMPLE_CLAMP, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_target_map_f32(0, 0, 1, 1, 1, 1, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_depth_output_f32(0, 0, 0, 0, 1, 1, 1, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_sample_stats_blocks(0) == 0 && banet_sample_stats_blocks(100000) > 0);
  EXPECT(banet_sample_stats_f32(0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_sample_stats_grad_f32(0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_sample_stats_grad_det_f32(0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_target_map_adjoint_f32(0, 0, 1, 1, 1, 1, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_dense_adjoint_workspace_bytes(0) == 0);
  EXPECT(banet_spd_solve_f32(0, 0, 0, 1, 134, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_spd_solve_f32(dummy, dummy, dummy, 0, 134, 0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_spd_solve_f32(dummy, dummy, dummy, 1, 8, 0) == BANET_ERR_UNSUPPORTED);      /* below two panels */
  EXPECT(banet_spd_solve_f32(dummy, dummy, dummy, 1, 400, 0) == BANET_ERR_UNSUPPORTED);    /* matrix does not fit the LDS */
  EXPECT(banet_profile_begin(0) == BANET_ERR_INVALID_ARG);
  EXPECT(banet_profile_end(0, 0, 0, 0, 0) == BANET_ERR_INVALID_ARG);
  printf("asan/ubsan driver: %d checks passed, no sanitizer report\n", checks);
  return 0;
}
EOC
RT=$(dirname $(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1))
/opt/rocm/lib/llvm/bin/clang -std=c99 -g -fsanitize=address,undefined -shared-libsan -I include $OUT/drive.c -o $OUT/drive -L$OUT -lbanet_hip_asan \
  -Wl,-rpath,$PWD/$OUT -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT
{
  echo "# tools/asan_capi.sh -- host side of libbanet_hip.so under AddressSanitizer + UBSan ($(date -u +%Y-%m-%d))"
  echo "# hipcc $(hipcc --version | grep -m1 -i 'hip version'), flags: -Xarch_host -fsanitize=address,undefined (halt on UB)"
  ASAN_OPTIONS=detect_leaks=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 $OUT/drive 2>&1
  echo "exit code $?"
} | tee profiles/r02_asan_capi.txt
