// C-ABI entry points of libbanet_hip.so (see include/banet_hip.h).  Argument checking,
// workspace carving and kernel enqueue only -- no allocation, no synchronisation.
#include "kernels.hpp"

namespace banet {

static bool aligned256(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 255u) == 0; }

static int check_level(const banet_level_t* lv) {
  if (!lv || !lv->src || !lv->tgt || !lv->depth) return BANET_ERR_INVALID_ARG;
  if (lv->variant < BANET_LEGACY_LM || lv->variant > BANET_BUNDLE) return BANET_ERR_INVALID_ARG;
  if (lv->K > 0 && !lv->basis) return BANET_ERR_INVALID_ARG;
  if ((lv->variant == BANET_BUNDLE) != (lv->K > 0)) return BANET_ERR_INVALID_ARG;
  if (lv->pairs < 0 || (lv->policy != BANET_POLICY_THROUGHPUT && lv->policy != BANET_POLICY_BATCH_INVARIANT)) return BANET_ERR_INVALID_ARG;
  if (lv->pairs > 1 && lv->variant != BANET_BUNDLE && lv->variant != BANET_BUNDLE_CAMERA) return BANET_ERR_INVALID_ARG;
  if (lv->dense) {
    if (!lv->intr || !(lv->scale > 0.f)) return BANET_ERR_INVALID_ARG;
  } else {
    if (!lv->rays || !lv->fx || !lv->fy || !lv->ox || !lv->oy) return BANET_ERR_INVALID_ARG;
  }
  return BANET_OK;
}

struct LevelWs {  // carve of the level workspace
  float* partials;
  float* AtA;
  float* Atb;
  float* absres;
  float* nvalid;
  LmCtl* ctl;
  float* mlp_y;   // [B] lambda-MLP outputs of the SYRK launch's role workgroups
  float* bigA;
  size_t total;
};

static LevelWs carve_level(const banet_level_t* lv, const AsmPlan& pl, void* ws) {
  LevelWs w;
  char* p = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(bytes, 256);
    return r;
  };
  w.partials = reinterpret_cast<float*>(take(pl.ws_bytes));
  w.AtA = reinterpret_cast<float*>(take((size_t)lv->B * pl.P * pl.P * sizeof(float)));
  w.Atb = reinterpret_cast<float*>(take((size_t)lv->B * pl.P * sizeof(float)));
  w.absres = reinterpret_cast<float*>(take((size_t)lv->B * lv->C * sizeof(float)));
  w.nvalid = reinterpret_cast<float*>(take((size_t)lv->B * sizeof(float)));
  w.ctl = reinterpret_cast<LmCtl*>(take((size_t)lv->B * sizeof(LmCtl)));
  w.mlp_y = reinterpret_cast<float*>(take((size_t)lv->B * sizeof(float)));
  const size_t big = solve_big_bytes(lv->B, pl.P, lv->C);
  w.bigA = big ? reinterpret_cast<float*>(take(big)) : nullptr;
  w.total = off;
  return w;
}

static SolveArgs make_solve_args(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base, const float* AtA,
                                 const float* Atb, const float* absres, const float* nvalid, const banet_state_t* st) {
  SolveArgs a;
  a.B = lv->B;
  a.N = lv->N;
  a.C = lv->C;
  a.K = lv->K;
  a.pairs = npairs(lv);
  a.P = 6 * a.pairs + lv->K;
  a.variant = lv->variant;
  a.l2_base = l2_base;
  a.max_iters = 0;
  a.use_mlp = lv->variant != BANET_LEGACY_FIXED;
  if (mlp) a.mlp = *mlp;
  else
    for (int i = 0; i < 5; ++i) a.mlp.w[i] = a.mlp.b[i] = nullptr;
  a.AtA = AtA;
  a.Atb = Atb;
  a.absres = absres;
  a.nvalid = nvalid;
  a.st = *st;
  a.ctl = nullptr;
  a.queue = nullptr;
  a.nqueue = 0;
  a.bigA = nullptr;
  a.mlp_y = nullptr;
  a.flags = lv->flags;
  banet_lm_params_default(&a.lm);
  return a;
}

static int check_state(const banet_level_t* lv, const banet_mlp_t* mlp, const banet_state_t* st) {
  if (!st || !st->R || !st->T || !st->iters || !st->ratio || !st->lambda_out || !st->delta) return BANET_ERR_INVALID_ARG;
  if (lv->K > 0 && !st->Wc) return BANET_ERR_INVALID_ARG;
  if (lv->variant != BANET_LEGACY_FIXED) {
    if (!mlp) return BANET_ERR_INVALID_ARG;
    for (int i = 0; i < 5; ++i)
      if (!mlp->w[i] || !mlp->b[i]) return BANET_ERR_INVALID_ARG;
  }
  return BANET_OK;
}

}  // namespace banet

using namespace banet;

extern "C" {

int banet_version(void) { return BANET_VERSION; }

const char* banet_error_string(int code) {
  switch (code) {
    case BANET_OK: return "ok";
    case BANET_ERR_INVALID_ARG: return "invalid argument";
    case BANET_ERR_WORKSPACE: return "workspace too small or misaligned";
    case BANET_ERR_UNSUPPORTED: return "shape not supported by the compiled kernel set";
    case BANET_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
  }
}

size_t banet_equation_construction_workspace_bytes(int B, int N, int C, int P) {
  EqPlan pl;
  if (plan_eq(B, N, C, P, &pl) != BANET_OK) return 0;
  return pl.partial_bytes;
}

int banet_equation_construction_f32(const float* J, const float* G, const float* d, float* AtA, float* Atb, int B,
                                    int N, int C, int P, void* ws, size_t ws_bytes, banet_stream_t stream) {
  if (!J || !G || !d || !AtA || !Atb) return BANET_ERR_INVALID_ARG;
  EqPlan pl;
  const int rc = plan_eq(B, N, C, P, &pl);
  if (rc != BANET_OK) return rc;
  if (!ws || ws_bytes < pl.partial_bytes || !aligned256(ws)) return BANET_ERR_WORKSPACE;
  return launch_eq(J, G, d, AtA, Atb, B, N, C, P, pl, static_cast<float*>(ws), static_cast<hipStream_t>(stream));
}

size_t banet_equation_construction_grad_workspace_bytes(int B, int N, int C, int P) {
  if (B <= 0 || N <= 0 || C <= 0 || P <= 0) return 0;
  return eq_grad_fast_ws_bytes(B, N, P);
}

int banet_equation_construction_grad_f32(const float* J, const float* G, const float* d, const float* g0,
                                         const float* g1, float* gJ, float* gG, float* gd, int B, int N, int C, int P,
                                         void* ws, size_t ws_bytes, banet_stream_t stream) {
  if (!J || !G || !d || !g0 || !g1 || !gJ || !gG || !gd) return BANET_ERR_INVALID_ARG;
  if (B <= 0 || N <= 0 || C <= 0 || P <= 0) return BANET_ERR_INVALID_ARG;
  // with a workspace (P <= 304): the matrix-pipe kernels of eqcon_grad.hip; without: the first-generation kernel
  const size_t need = eq_grad_fast_ws_bytes(B, N, P);
  if (need > 0 && ws != nullptr && ws_bytes >= need && aligned256(ws))
    return launch_eq_grad_fast(J, G, d, g0, g1, gJ, gG, gd, B, N, C, P, ws, static_cast<hipStream_t>(stream));
  return launch_eq_grad(J, G, d, g0, g1, gJ, gG, gd, B, N, C, P, static_cast<hipStream_t>(stream));
}

size_t banet_ba_assemble_workspace_bytes(const banet_level_t* lv) {
  AsmPlan pl;
  if (plan_assemble(lv, &pl) != BANET_OK) return 0;
  return align_up(pl.ws_bytes, 256);
}

int banet_ba_assemble_f32(const banet_level_t* lv, const float* R, const float* T, const float* Wc, float* AtA,
                          float* Atb, float* absres, float* nvalid, void* ws, size_t ws_bytes, banet_stream_t stream) {
  int rc = check_level(lv);
  if (rc != BANET_OK) return rc;
  if (!R || !T || !AtA || !Atb || !absres || !nvalid || (lv->K > 0 && !Wc)) return BANET_ERR_INVALID_ARG;
  AsmPlan pl;
  rc = plan_assemble(lv, &pl);
  if (rc != BANET_OK) return rc;
  if (!ws || ws_bytes < pl.ws_bytes || !aligned256(ws)) return BANET_ERR_WORKSPACE;
  return launch_assemble(lv, pl, R, T, Wc, nullptr, 0, ws, AtA, Atb, absres, nvalid,
                         static_cast<hipStream_t>(stream), true, nullptr, nullptr, nullptr, pl.s.f16_standalone ? 0 : -1);
}

int banet_ba_assemble_mask_f32(const banet_level_t* lv, const float* R, const float* T, const float* Wc, float* AtA,
                               float* Atb, float* absres, float* nvalid, unsigned char* mask_out, void* ws, size_t ws_bytes,
                               banet_stream_t stream) {
  int rc = check_level(lv);
  if (rc != BANET_OK) return rc;
  if (!R || !T || !AtA || !Atb || !absres || !nvalid || !mask_out || (lv->K > 0 && !Wc)) return BANET_ERR_INVALID_ARG;
  AsmPlan pl;
  rc = plan_assemble(lv, &pl);
  if (rc != BANET_OK) return rc;
  if (!ws || ws_bytes < pl.ws_bytes || !aligned256(ws)) return BANET_ERR_WORKSPACE;
  return launch_assemble(lv, pl, R, T, Wc, nullptr, 0, ws, AtA, Atb, absres, nvalid, static_cast<hipStream_t>(stream), true,
                         nullptr, nullptr, mask_out, pl.s.f16_standalone ? 0 : -1);
}

int banet_ba_solve_update_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base, const float* AtA,
                              const float* Atb, const float* absres, const float* nvalid, banet_state_t* st,
                              banet_stream_t stream) {
  if (!lv || !AtA || !Atb || !absres || !nvalid) return BANET_ERR_INVALID_ARG;
  if (lv->variant < BANET_LEGACY_LM || lv->variant > BANET_BUNDLE || lv->B <= 0 || lv->N <= 0 || lv->C <= 0 || lv->K < 0 ||
      lv->pairs < 0)
    return BANET_ERR_INVALID_ARG;
  const int rc = check_state(lv, mlp, st);
  if (rc != BANET_OK) return rc;
  SolveArgs a = make_solve_args(lv, mlp, l2_base, AtA, Atb, absres, nvalid, st);
  return launch_solve(a, static_cast<hipStream_t>(stream));
}

size_t banet_ba_solve_update_workspace_bytes(const banet_level_t* lv) {
  if (!lv || lv->B <= 0 || lv->C <= 0 || lv->K < 0 || lv->pairs < 0) return 0;
  return align_up(solve_big_bytes(lv->B, 6 * npairs(lv) + lv->K, lv->C), 256);
}

int banet_ba_solve_update_ws_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base, const float* AtA,
                                 const float* Atb, const float* absres, const float* nvalid, banet_state_t* st, void* ws,
                                 size_t ws_bytes, banet_stream_t stream) {
  if (!lv || !AtA || !Atb || !absres || !nvalid) return BANET_ERR_INVALID_ARG;
  if (lv->variant < BANET_LEGACY_LM || lv->variant > BANET_BUNDLE || lv->B <= 0 || lv->N <= 0 || lv->C <= 0 || lv->K < 0 ||
      lv->pairs < 0)
    return BANET_ERR_INVALID_ARG;
  const int rc = check_state(lv, mlp, st);
  if (rc != BANET_OK) return rc;
  SolveArgs a = make_solve_args(lv, mlp, l2_base, AtA, Atb, absres, nvalid, st);
  const size_t need = banet_ba_solve_update_workspace_bytes(lv);
  if (need > 0) {
    if (!ws || ws_bytes < need || !aligned256(ws)) return BANET_ERR_WORKSPACE;
    a.bigA = static_cast<float*>(ws);
  }
  return launch_solve(a, static_cast<hipStream_t>(stream));
}

size_t banet_lm_level_workspace_bytes(const banet_level_t* lv) {
  AsmPlan pl;
  if (plan_assemble(lv, &pl) != BANET_OK) return 0;
  return carve_level(lv, pl, nullptr).total;
}

void banet_lm_params_default(banet_lm_params_t* p) {
  if (!p) return;
  p->angle_change = (float)(0.002 * (3.14 / 180.0));   // legacy/ba.py:6 (3.14, not pi)
  p->translation_change = 0.0002f;                      // legacy/ba.py:7
  p->residual_ratio = 1.0f;                             // legacy/ba.py:8
  p->solver = BANET_SOLVER_QR;                          // legacy/ba.py:9
}

constexpr int kRoleSmallLevel = 19200;   // pixels: levels whose SYRK launch is latency-bound (see the role condition below)

int banet_lm_level_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base, int max_iters,
                       int early_termination, banet_state_t* st, void* ws, size_t ws_bytes, banet_stream_t stream) {
  return banet_lm_level_ex_f32(lv, mlp, l2_base, max_iters, early_termination, nullptr, st, ws, ws_bytes, stream);
}

int banet_lm_level_ex_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base, int max_iters,
                          int early_termination, const banet_lm_params_t* params, banet_state_t* st, void* ws,
                          size_t ws_bytes, banet_stream_t stream) {
  int rc = check_level(lv);
  if (rc != BANET_OK) return rc;
  rc = check_state(lv, mlp, st);
  if (rc != BANET_OK) return rc;
  if (max_iters < 0) return BANET_ERR_INVALID_ARG;
  AsmPlan pl;
  rc = plan_assemble(lv, &pl);
  if (rc != BANET_OK) return rc;
  if (!ws || !aligned256(ws)) return BANET_ERR_WORKSPACE;
  LevelWs w = carve_level(lv, pl, ws);
  if (ws_bytes < w.total) return BANET_ERR_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  SolveArgs a = make_solve_args(lv, mlp, l2_base, w.AtA, w.Atb, w.absres, w.nvalid, st);
  a.max_iters = max_iters;
  a.bigA = w.bigA;
  if (params) {
    if (params->solver != BANET_SOLVER_QR && params->solver != BANET_SOLVER_INVERSE) return BANET_ERR_INVALID_ARG;
    if (!(params->angle_change >= 0.f) || !(params->translation_change >= 0.f) || !(params->residual_ratio > 0.f))
      return BANET_ERR_INVALID_ARG;   // also rejects NaN
    a.lm = *params;
  }
  const bool lm = early_termination && lv->variant == BANET_LEGACY_LM;
  if (lm) {
    // device-side loop control: max_iters + 1 evaluation rounds; the last one only runs the
    // pending accept/reject test (legacy/ba.py:304-345), see solve.hip
    launch_ctl_init(w.ctl, st->iters, lv->B, s);
    a.ctl = w.ctl;
    const int stride = (int)(sizeof(LmCtl) / sizeof(int32_t));
    for (int it = 0; it <= max_iters; ++it) {
      rc = launch_assemble(lv, pl, st->R, st->T, st->Wc, &w.ctl->active, stride, w.partials, w.AtA, w.Atb, w.absres,
                           w.nvalid, s);
      if (rc != BANET_OK) return rc;
      {
        RangeScope r("solve", lv->N);
        rc = launch_solve(a, s);
      }
      if (rc != BANET_OK) return rc;
    }
  } else {
    launch_zero_iters(st->iters, lv->B, s);
    // the tile queue is reset once here; afterwards every solve kernel leaves it zeroed for the next gather
    a.queue = assemble_queue(pl, w.partials);
    a.nqueue = 8 * npairs(lv);
    if (a.queue) launch_zero_iters(a.queue, lv->B * a.nqueue, s);   // a kernel, not hipMemsetAsync: see prepare_gather
    // bundle levels whose SYRK is ba_syrk_bf16x6_kernel: the lambda MLP runs as a role workgroup of the SYRK launch, off the
    // solve kernel's critical path (C <= 256: the role's LDS scratch)
    // ... and coarse levels at any batch (N <= kRoleSmallLevel pixels: their SYRK is a latency chain of 1-3 steps per wave, so a
    // window's 7 workgroups take what 8 take, and the solve kernel loses the 19 us MLP: 40x30 .. 160x120 x 32 windows).
    // Small batches only (B <= 8) otherwise: the SYRK kernel runs one workgroup per CU (512 registers per wave), so the role workgroups
    // need CUs of their own -- at B = 32 (8 + 1 workgroups per window = 288 > 256 CUs) a second round of workgroups doubled
    // the SYRK time (640x480 x 32: 1326 -> 2457 us); with B <= 8 one SYRK workgroup per window is given up where needed.
    const int Bsel = selection_batch(lv);   // (the role changes Gs, i.e. the summation split: decided like every other selection)
    const bool role = lv->variant == BANET_BUNDLE && mlp != nullptr && a.use_mlp && syrk_runs_mlp_role(pl.s) && lv->C <= 256 &&
                      (lv->C & 3) == 0 && ((long long)Bsel * (pl.s.Gs + 1) <= num_cus() || (Bsel <= 8 && pl.s.Gs >= 16) ||
                       (lv->N <= kRoleSmallLevel && pl.s.Gs >= 4)) &&
                      !(lv->flags & 32768);   // flags bit 15: MLP inside the solve kernel (A/B)
    if (role) {
      a.mlp_y = w.mlp_y;
      if ((long long)Bsel * (pl.s.Gs + 1) > num_cus()) pl.s.Gs -= 1;
    }
    for (int it = 0; it < max_iters; ++it) {
      rc = launch_assemble(lv, pl, st->R, st->T, st->Wc, nullptr, 0, w.partials, w.AtA, w.Atb, w.absres, w.nvalid, s,
                           a.queue == nullptr, role ? mlp : nullptr, role ? w.mlp_y : nullptr, nullptr,
                           // fp16 two-piece SYRK: the level's first pass runs the exact form and leaves the basis column maxima
                           // on the way (no extra pass over the basis); a one-iteration call computes them up front instead
                           pl.s.f16 ? (it == 0 ? (max_iters > 1 ? 2 : 0) : 1) : -1);
      if (rc != BANET_OK) return rc;
      {
        RangeScope r("solve", lv->N);
        rc = launch_solve(a, s);
      }
      if (rc != BANET_OK) return rc;
    }
  }
  return BANET_OK;
}

int banet_resample_f32(const float* data, const float* warp, float* out, int B, int N, int C, int H, int W, int mode,
                       banet_stream_t stream) {
  if (!data || !warp || !out) return BANET_ERR_INVALID_ARG;
  return launch_resample(data, warp, out, B, N, C, H, W, mode, static_cast<hipStream_t>(stream));
}

int banet_target_map_f32(const float* img, float* out, int B, int H, int W, int C, banet_stream_t stream) {
  if (!img || !out) return BANET_ERR_INVALID_ARG;
  return launch_target_map(img, out, B, H, W, C, static_cast<hipStream_t>(stream));
}

int banet_depth_output_f32(const float* init_depth, const float* basis, const float* Wc, float* out, int B, int N, int K,
                           banet_stream_t stream) {
  if (!init_depth || !basis || !Wc || !out) return BANET_ERR_INVALID_ARG;
  return launch_depth_output(init_depth, basis, Wc, out, B, N, K, static_cast<hipStream_t>(stream));
}

int banet_sample_stats_blocks(int N) { return N > 0 ? sample_stats_blocks(N) : 0; }

static bool sstats_shape_ok(int B, int N, int C, int H, int W) {
  return B > 0 && N > 0 && C > 0 && H > 0 && W > 0 && (unsigned long long)B * H * W * 3ull * C < (1ull << 32);
}

int banet_sample_stats_f32(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C, int H,
                           int W, float* stats, float* absd_part, banet_stream_t stream) {
  if (!conv1 || !conv2 || !px || !py || !stats || !absd_part) return BANET_ERR_INVALID_ARG;
  if (!sstats_shape_ok(B, N, C, H, W)) return BANET_ERR_INVALID_ARG;
  return launch_sample_stats(conv1, conv2, px, py, B, N, C, H, W, stats, absd_part, static_cast<hipStream_t>(stream));
}

int banet_sample_stats_grad_f32(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C,
                                int H, int W, const float* dstats, const float* dabs, float* dconv1, float* dconv2, float* dpos,
                                banet_stream_t stream) {
  if (!conv1 || !conv2 || !px || !py || !dstats || !dabs || !dconv1 || !dconv2 || !dpos) return BANET_ERR_INVALID_ARG;
  if (!sstats_shape_ok(B, N, C, H, W)) return BANET_ERR_INVALID_ARG;
  return launch_sample_stats_grad(conv1, conv2, px, py, B, N, C, H, W, dstats, dabs, dconv1, dconv2, dpos,
                                  static_cast<hipStream_t>(stream));
}

int banet_spd_solve_f32(const float* A, const float* rhs, float* x, int B, int P, banet_stream_t stream) {
  if (!A || !rhs || !x || B <= 0 || P <= 0) return BANET_ERR_INVALID_ARG;
  return launch_spd_solve(A, rhs, x, B, P, static_cast<hipStream_t>(stream));
}

size_t banet_sample_stats_grad_workspace_bytes(int B, int N, int C, int H, int W) {
  if (!sstats_shape_ok(B, N, C, H, W)) return 0;
  return sample_stats_grad_det_workspace_bytes(B, N, C, H, W);
}

int banet_sample_stats_grad_det_f32(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C,
                                    int H, int W, const float* dstats, const float* dabs, float* dconv1, float* dconv2,
                                    float* dpos, void* ws, size_t ws_bytes, banet_stream_t stream) {
  if (!conv1 || !conv2 || !px || !py || !dstats || !dabs || !dconv1 || !dconv2 || !dpos || !ws) return BANET_ERR_INVALID_ARG;
  if (!sstats_shape_ok(B, N, C, H, W)) return BANET_ERR_INVALID_ARG;
  const size_t need = sample_stats_grad_det_workspace_bytes(B, N, C, H, W);
  if (need == 0) return BANET_ERR_UNSUPPORTED;
  if (ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 255) != 0) return BANET_ERR_WORKSPACE;
  return launch_sample_stats_grad_det(conv1, conv2, px, py, B, N, C, H, W, dstats, dabs, dconv1, dconv2, dpos, ws,
                                      static_cast<hipStream_t>(stream));
}

size_t banet_dense_adjoint_workspace_bytes(const banet_level_t* lv) {
  if (!lv || lv->B <= 0 || lv->N <= 0) return 0;
  return dense_adjoint_workspace_bytes(lv);
}

size_t banet_dense_adjoint_workspace_bytes_ex(const banet_level_t* lv, int flags) {
  if (!lv || lv->B <= 0 || lv->N <= 0) return 0;
  if (flags & ~(BANET_ADJOINT_OVERWRITE | BANET_ADJOINT_OVERWRITE_MAP | BANET_ADJOINT_FOLD_TARGET | BANET_ADJOINT_REUSE_DEPTH_SEED | BANET_ADJOINT_TILE_SHAPE(15))) return 0;
  return dense_adjoint_workspace_bytes(lv, flags);
}

int banet_dense_adjoint_ex_f32(const banet_level_t* lv, const float* R, const float* T, const float* Wc, const float* gAtA,
                               const float* gAtb, const float* gabs, float* dsrc, float* dmap3, float* ddepth, float* dbasis,
                               float* dpose, int flags, void* ws, size_t ws_bytes, banet_stream_t stream) {
  if (!lv || !R || !T || !gAtA || !gAtb || !gabs || !dsrc || !dmap3 || !ddepth || !dpose || !ws) return BANET_ERR_INVALID_ARG;
  if (lv->K > 0 && (!Wc || !dbasis || !lv->basis)) return BANET_ERR_INVALID_ARG;     // K = 0 (pose only): no coefficient / basis tensors
  if (lv->B <= 0 || lv->N <= 0 || !lv->src || !lv->tgt || !lv->depth) return BANET_ERR_INVALID_ARG;
  if (lv->dense ? !lv->intr : (!lv->rays || !lv->fx || !lv->fy || !lv->ox || !lv->oy)) return BANET_ERR_INVALID_ARG;
  if (flags & ~(BANET_ADJOINT_OVERWRITE | BANET_ADJOINT_OVERWRITE_MAP | BANET_ADJOINT_FOLD_TARGET | BANET_ADJOINT_REUSE_DEPTH_SEED | BANET_ADJOINT_TILE_SHAPE(15))) return BANET_ERR_INVALID_ARG;
  const size_t need = dense_adjoint_workspace_bytes(lv, flags);
  if (need == 0) return BANET_ERR_UNSUPPORTED;
  if (ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 255) != 0) return BANET_ERR_WORKSPACE;
  return launch_dense_adjoint(lv, R, T, Wc, gAtA, gAtb, gabs, dsrc, dmap3, ddepth, dbasis, dpose, flags, ws,
                              static_cast<hipStream_t>(stream));
}

int banet_dense_adjoint_f32(const banet_level_t* lv, const float* R, const float* T, const float* Wc, const float* gAtA,
                            const float* gAtb, const float* gabs, float* dsrc, float* dmap3, float* ddepth, float* dbasis,
                            float* dpose, void* ws, size_t ws_bytes, banet_stream_t stream) {
  return banet_dense_adjoint_ex_f32(lv, R, T, Wc, gAtA, gAtb, gabs, dsrc, dmap3, ddepth, dbasis, dpose, 0, ws, ws_bytes, stream);
}

int banet_target_map_adjoint_ex_f32(const float* dmap3, float* dimg, int B, int H, int W, int C, int flags, banet_stream_t stream) {
  if (!dmap3 || !dimg || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (flags & ~BANET_ADJOINT_OVERWRITE)) return BANET_ERR_INVALID_ARG;
  return launch_target_map_adjoint(dmap3, dimg, B, H, W, C, flags & BANET_ADJOINT_OVERWRITE, static_cast<hipStream_t>(stream));
}

int banet_target_map_adjoint_f32(const float* dmap3, float* dimg, int B, int H, int W, int C, banet_stream_t stream) {
  return banet_target_map_adjoint_ex_f32(dmap3, dimg, B, H, W, C, 0, stream);
}

size_t banet_small_step_adjoint_workspace_bytes(int variant, int B, int N, int C, int K, int pairs) {
  return small_step_workspace_bytes(variant, B, N, C, K, pairs);
}

int banet_small_step_adjoint_f32(int variant, int B, int N, int C, int K, int pairs, float l2_regularizer_base, const banet_mlp_t* mlp,
                                 const float* AtA, const float* Atb, const float* absres, const float* delta, const float* R,
                                 const float* T, const float* gR, const float* gT, const float* gW, float* gAtA, float* gAtb, float* gabs,
                                 float* dR, float* dT, const banet_mlp_t* gmlp, void* ws, size_t ws_bytes, banet_stream_t stream) {
  if (!mlp || !gmlp || !AtA || !Atb || !absres || !delta || !R || !T || !gR || !gT || !gAtA || !gAtb || !gabs || !dR || !dT || !ws)
    return BANET_ERR_INVALID_ARG;
  if (K > 0 && !gW) return BANET_ERR_INVALID_ARG;
  for (int i = 0; i < 5; ++i)
    if (!mlp->w[i] || !mlp->b[i] || !gmlp->w[i] || !gmlp->b[i]) return BANET_ERR_INVALID_ARG;
  if (B <= 0 || N <= 0 || C <= 0 || K < 0 || pairs < 1) return BANET_ERR_INVALID_ARG;
  const size_t need = small_step_workspace_bytes(variant, B, N, C, K, pairs);
  if (need == 0) return BANET_ERR_UNSUPPORTED;
  if (ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 255) != 0) return BANET_ERR_WORKSPACE;
  return launch_small_step_adjoint(variant, B, N, C, K, pairs, l2_regularizer_base, mlp, AtA, Atb, absres, delta, R, T, gR, gT, gW, gAtA,
                                   gAtb, gabs, dR, dT, gmlp, ws, static_cast<hipStream_t>(stream));
}

#include "build_id.h"   // generated by build.sh: BANET_BUILD_ID
const char* banet_build_id(void) { return BANET_BUILD_ID; }

int banet_profile_ranges(int enable) { return profile_ranges(enable); }

int banet_gather_selection(const banet_level_t* lv) {
  GatherPlan pl;
  const int rc = plan_gather(lv, &pl);
  if (rc != BANET_OK) return rc;
  return pl.quad ? 4 : pl.strip ? 3 : pl.patch ? 2 : pl.c128 ? 1 : 0;
}

int banet_syrk_selection(const banet_level_t* lv) {
  AsmPlan pl;
  const int rc = plan_assemble(lv, &pl);
  if (rc != BANET_OK) return rc;
  if (lv->K == 0) return -1000;
  return pl.s.f16 ? 4 : pl.s.direct;
}

int banet_profile_begin(int max_launches) { return profile_begin(max_launches); }

int banet_profile_end(int max_tags, int32_t* tag_points, int32_t* tag_launches, double* tag_ms, int32_t* ntags) {
  return profile_end(max_tags, tag_points, tag_launches, tag_ms, ntags);
}

}  // extern "C"
