// Host-visible plans / argument blocks shared by the translation units of libbanet_hip.so.
#pragma once
#include "common.hpp"

namespace banet {

constexpr int kGHdr = 32;       // gather partial header: 21 H_cc + 6 Atb_c + nvalid (+pad), then C x sum|d|
constexpr int kUStrideS = 8;    // per-pixel record: u0..u5, s, r

inline int npairs(const banet_level_t* lv) { return lv->pairs > 1 ? lv->pairs : 1; }
// the batch size every kernel / arithmetic-form / partial-row decision is taken from (banet_level_t.policy): the launch's own, or
// -- BANET_POLICY_BATCH_INVARIANT -- the canonical one, so that a window's bits do not depend on who shares its launch
inline int selection_batch(const banet_level_t* lv) { return lv->policy == BANET_POLICY_BATCH_INVARIANT ? BANET_CANONICAL_BATCH : lv->B; }

// compute units of the current device (hipDeviceAttributeMultiprocessorCount, queried once per device; 256 = MI355X when no
// device is visible, e.g. the host-only plan / workspace arithmetic of the CPU tests).  The launch plans size their grids by it.
int num_cus();

// ---- gather.hip ------------------------------------------------------------------------
struct GatherPlan {
  int G, tiles, tiles_x, tiles_y, groups, pstride;
  int c128;     // 1: ba_gather128_kernel (dynamic tile queue, one partial row per tile)
  int patch;    // 1: ba_gather128p_kernel (same interface; taps from wave-private LDS patches) for large levels
  int strip;    // ba_gather128s_kernel (work items = 16-pixel-wide strip segments, rolling LDS window): pixel rows per segment
                //    (32 or 16; 0 = another kernel); tiles_x / tiles_y / tiles then count segments
  int quad;     // 1: ba_gather128q_kernel (work items = 4x4 pixel blocks, one step per item): latency-bound launches; tiles_x / tiles_y / tiles count items
  int strip_fp; // strip kernel, multi-frame windows: a workgroup = `pairs` waves on one segment, wave p against target frame p
  int rows;     // partial rows per window written by the gather kernel (tiles or G)
  int frows;    // rows per window handed to ba_reduce2_kernel (after ba_fold_kernel when rows > kFoldRows)
  int nbands;   // tile-queue bands (8 = one per XCD)
  int pairloop; // patch kernel: target frames looped over inside a tile (grid y = windows)
  int qshift;   // 2: quarter-tile work items (levels with fewer tiles than resident waves)
  int tile_pts; // generic kernel, sparse points: points per wave item (64; 16 on latency-bound launches)
  size_t off_fold, off_queue;   // inside the partial region
  size_t partial_bytes, rec_bytes;
};
constexpr int kFoldRows = 64;
int plan_gather(const banet_level_t* lv, GatherPlan* pl);
// prepare (reset the tile queue) -> launch (the gather kernel alone: this is what the profiler times)
// -> finish (fold the tile partials); `reduced` returns the rows ba_reduce2_kernel should read
void prepare_gather(const banet_level_t* lv, const GatherPlan& pl, float* partials, hipStream_t s);
int launch_gather(const banet_level_t* lv, const GatherPlan& pl, const float* R, const float* T, const float* Wc,
                  const int32_t* active, int active_stride, float* rec, float* partials, hipStream_t s,
                  unsigned char* mask_out = nullptr);
const float* finish_gather(const banet_level_t* lv, const GatherPlan& pl, const int32_t* active, int active_stride,
                           float* partials, hipStream_t s);

// ---- syrk.hip --------------------------------------------------------------------------
struct MlpRole {        // one extra workgroup per window of the SYRK launch evaluates the lambda MLP (mlp.hpp); y == nullptr: off
  const float* gpart;   // folded gather partial rows [B * pairs][grows][gstride]
  int grows, gstride;
  banet_mlp_t mlp;
  int C, pairs;
  float Nf;             // residual rows per window (N * pairs)
  float* y;             // [B] MLP output
};
struct SyrkPlan {
  int Gs, tiles, pstride, nb;
  int x3;       // ba_syrk_bf16x6_kernel, opt-in: three products instead of six (flags bit 29)
  int direct;   // 0: the LDS-tiled kernel, 1: ba_syrk_direct_kernel (fp32 MFMA, A/B), 2: ba_syrk_bf16x6_kernel (K = 64 / 128, <= 4 frames),
                // 3: syrk_wide.hip jobs (K = 256, or K = 128 with more than 4 target frames)
  int f16;      // ba_syrk_bf16x6_kernel: the fp16 two-piece form is eligible (plan_syrk); f16_standalone: also in a single assembly pass
  int f16_standalone;
  size_t off_colmax, off_recmax;   // f16: [B][K] basis column maxima, [B][32][2] record maxima, inside the partial buffer
  size_t off_aux;        // direct == 3: per-pixel (s, r) sums over the frames, inside the partial buffer
  size_t partial_bytes;
};
size_t syrk_wide_aux_bytes(int B, int N, int pairs);
int launch_syrk_wide(const float* basis, const float* rec, int B, int N, int K, int pairs, int Gs, int pstride,
                     const int32_t* active, int active_stride, float* partials, float* aux, hipStream_t s,
                     const float* colmax = nullptr, const float* recmax = nullptr);   // both given: the fp16 two-piece form
int plan_syrk(int B, int Bsel, int N, int K, int pairs, int dbg, SyrkPlan* pl);   // Bsel: the batch the decisions are taken from (selection_batch)
int launch_syrk(const float* basis, const float* rec, int B, int N, int K, int pairs, const SyrkPlan& pl,
                const int32_t* active, int active_stride, float* partials, hipStream_t s, const MlpRole* mr = nullptr,
                int f16_stats = -1);   // f16_stats: -1 = exact bf16 form; 0 / 1 = fp16 two-piece form (pl.f16), basis column maxima to compute / in
                                       // place; 2 = exact bf16 form that also leaves the column maxima (first iteration of a level)
inline bool syrk_runs_mlp_role(const SyrkPlan& pl) { return pl.direct == 2; }   // ba_syrk_bf16x6_kernel only
void launch_reduce2(const float* gpart, int Gg, int gstride, const float* spart, int Gs, int sstride,
                    const int32_t* active, int active_stride, int B, int K, int C, int pairs, float* AtA, float* Atb,
                    float* absres, float* nvalid, hipStream_t s);

// ---- assemble.hip: one assembly pass = gather + syrk + reduce ------------------------------
struct AsmPlan {
  GatherPlan g;
  SyrkPlan s;
  int P;
  size_t ws_bytes;      // gather partials + records + syrk partials
  size_t off_rec, off_spart;
};
int plan_assemble(const banet_level_t* lv, AsmPlan* pl);
int launch_assemble(const banet_level_t* lv, const AsmPlan& pl, const float* R, const float* T, const float* Wc,
                    const int32_t* active, int active_stride, void* ws, float* AtA, float* Atb, float* absres,
                    float* nvalid, hipStream_t s, bool reset_queue = true, const banet_mlp_t* role_mlp = nullptr,
                    float* role_y = nullptr, unsigned char* mask_out = nullptr, int f16_stats = -1);
int* assemble_queue(const AsmPlan& pl, void* ws);   // the gather's tile-queue heads inside the workspace (or nullptr)
int profile_ranges(int enable);             // roctx ranges around the launches (banet_profile_ranges)
struct RangeScope {                         // pushes "banet.<role>[ N=<n>]" when ranges are on
  bool on;
  RangeScope(const char* role, int n = -1);
  ~RangeScope();
};
int profile_begin(int max_launches);
int profile_end(int max_tags, int32_t* tag_points, int32_t* tag_launches, double* tag_ms, int32_t* ntags);

// ---- prep.hip --------------------------------------------------------------------------
int launch_resample(const float* data, const float* warp, float* out, int B, int N, int C, int H, int W, int mode,
                    hipStream_t s);
int launch_target_map(const float* img, float* out, int B, int H, int W, int C, hipStream_t s);
int launch_depth_output(const float* init, const float* basis, const float* Wc, float* out, int B, int N, int K,
                        hipStream_t s);

// ---- eqcon.hip -------------------------------------------------------------------------
void launch_reduce(const float* partials, int B, int G, int pstride, int P, float* AtA, float* Atb, hipStream_t s);
struct EqPlan {
  int Gr, tiles, pstride, nb;
  int fast;            // 1: eqcon_syrk.hip (P <= 144): per-pixel records + W^T W on the bf16 pipe
  size_t off_rec;      // fast: the records [B][N][8] inside the workspace
  size_t partial_bytes;
};
size_t eq_syrk_record_bytes(int B, int N);
int launch_eq_syrk(const float* J, const float* G, const float* d, int B, int N, int C, int P, int nb, int Gr, int pstride,
                   float* partials, float* rec, hipStream_t s);
int plan_eq(int B, int N, int C, int P, EqPlan* pl);
int launch_eq(const float* J, const float* G, const float* d, float* AtA, float* Atb, int B, int N, int C, int P,
              const EqPlan& pl, float* partials, hipStream_t s);
size_t eq_grad_fast_ws_bytes(int B, int N, int P);   // eqcon_grad.hip: 0 = no fast path for this shape
int launch_eq_grad_fast(const float* J, const float* G, const float* d, const float* g0, const float* g1, float* gJ, float* gG,
                        float* gd, int B, int N, int C, int P, void* ws, hipStream_t s);
int launch_eq_grad(const float* J, const float* G, const float* d, const float* g0, const float* g1, float* gJ,
                   float* gG, float* gd, int B, int N, int C, int P, hipStream_t s);

// ---- solve.hip -------------------------------------------------------------------------
struct LmCtl {  // per-window loop state of the legacy early-termination LM (device memory)
  int32_t active;
  int32_t pending;
  float avg_prev;
  float uw, ut;
  float Rprev[9];
  float Tprev[3];
  int32_t pad_[3];
};
static_assert(sizeof(LmCtl) == 80, "LmCtl layout");

struct SolveArgs {
  int B, N, C, K, P, variant, pairs;
  float l2_base;
  int max_iters;  // only used with ctl
  banet_mlp_t mlp;
  int use_mlp;
  const float* AtA;
  const float* Atb;
  const float* absres;
  const float* nvalid;
  banet_state_t st;
  LmCtl* ctl;  // nullptr: fixed-count mode (every call performs one update)
  float* bigA; // workspace for the normal matrix when it does not fit in LDS (solve_big_bytes), else nullptr
  int* queue;  // LM loop: the next gather's tile-queue heads, zeroed by this kernel (saves a memset per iteration)
  int nqueue;  // words per window
  const float* mlp_y;   // [B] lambda-MLP outputs precomputed by the SYRK launch's role workgroups, or nullptr
  banet_lm_params_t lm;   // run-time LM configuration (legacy/ba.py:5-9)
  int flags;              // banet_level_t.flags (development switches: bit 23 = blocked LDL^T only, no conjugate gradients)
};
int launch_solve(const SolveArgs& a, hipStream_t s);
int launch_spd_solve(const float* A, const float* rhs, float* x, int B, int P, hipStream_t s);   // 32 <= P, matrix in LDS
bool spd_solve_fits(int P);     // launch_spd_solve accepts this size
size_t solve_big_bytes(int B, int P, int C);
void launch_ctl_init(LmCtl* ctl, int32_t* iters, int B, hipStream_t s);
void launch_zero_iters(int32_t* iters, int B, hipStream_t s);

// ---- sstats.hip: per-pixel sampling statistics and their adjoint (differentiable layer support) ----
int sample_stats_blocks(int N);
int launch_sample_stats(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C, int H,
                        int W, float* stats, float* absd_part, hipStream_t s);
int launch_sample_stats_grad(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C,
                             int H, int W, const float* dstats, const float* dabs, float* dconv1, float* dconv2, float* dpos,
                             hipStream_t s);

// ---- adjoint.hip: backward of the fused dense bundle assembly ----
size_t dense_adjoint_workspace_bytes(const banet_level_t* lv, int flags = 0);   // flags: BANET_ADJOINT_FOLD_TARGET changes the layout
int launch_dense_adjoint(const banet_level_t* lv, const float* R, const float* T, const float* Wc, const float* gAtA,
                         const float* gAtb, const float* gabs, float* dsrc, float* dmap3, float* ddepth, float* dbasis,
                         float* dpose, int flags, void* ws, hipStream_t s);
size_t sample_stats_grad_det_workspace_bytes(int B, int N, int C, int H, int W);
int launch_sample_stats_grad_det(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C,
                                 int H, int W, const float* dstats, const float* dabs, float* dconv1, float* dconv2, float* dpos,
                                 void* ws, hipStream_t s);
int launch_target_map_adjoint(const float* dmap3, float* dimg, int B, int H, int W, int C, int overwrite, hipStream_t s);

// ---- smallstep.hip: backward of the small part of an iteration (lambda MLP, damping, solve, SE(3) / W update) ----
bool small_step_supported(int variant, int B, int N, int C, int K, int pairs);
size_t small_step_workspace_bytes(int variant, int B, int N, int C, int K, int pairs);
int launch_small_step_adjoint(int variant, int B, int N, int C, int K, int pairs, float l2_base, const banet_mlp_t* mlp, const float* AtA,
                              const float* Atb, const float* absres, const float* delta, const float* R, const float* T, const float* gR,
                              const float* gT, const float* gW, float* gAtA, float* gAtb, float* gabs, float* dR, float* dT,
                              const banet_mlp_t* gmlp, void* ws, hipStream_t s);

}  // namespace banet
