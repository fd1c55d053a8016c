/* banet_hip.h -- C ABI of libbanet_hip.so: the MI355X (gfx950) bundle-adjustment hot path.
 *
 * This is the drop-in boundary for the one native component of frobelbest/BANet, the
 * TensorFlow custom-op library `utils.so` (reference: utils.cu, loaded by
 * bundlenet.py:76-82 and legacy/ba.py:11-13), plus fused entry points that replace the
 * TF-graph portion of the same path (bundlenet.py:122-278, legacy/ba.py:148-345).
 *
 * Conventions (all entry points):
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer owned by the caller;
 *   - row-major contiguous float32 tensors with the reference's axis order;
 *   - asynchronous: work is enqueued on the caller's hipStream_t; nothing synchronises,
 *     allocates or frees; scratch comes from the caller-supplied workspace `ws`
 *     (size from the matching *_workspace_bytes query; 256-byte aligned);
 *   - stateless and re-entrant: no globals (the reference keeps per-GPU static scratch
 *     sized by the first call, utils.cu:214-216,259-264 -- not reproduced);
 *   - return value: BANET_OK or a negative BANET_ERR_* code; never throws.
 *   - deterministic: reductions use fixed-order partials, no float atomics.
 */
#ifndef BANET_HIP_H_
#define BANET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef BANET_STREAM_T
#define BANET_STREAM_T
typedef void* banet_stream_t; /* a hipStream_t (NULL = default stream) */
#endif

#define BANET_VERSION 150 /* 0.1.5: BANET_ADJOINT_FOLD_TARGET, banet_small_step_adjoint_f32; 0.1.4: banet_dense_adjoint_f32 / banet_target_map_adjoint_f32; 0.1.3: banet_lm_params_t / banet_lm_level_ex_f32 (0.1.2: banet_sample_stats[_grad]_f32; 0.1.1: banet_level_t.pairs) */

enum {
  BANET_OK = 0,
  BANET_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, bad enum        */
  BANET_ERR_WORKSPACE = -2,   /* ws too small or misaligned                       */
  BANET_ERR_UNSUPPORTED = -3, /* shape outside the compiled kernel set (see docs) */
  BANET_ERR_LAUNCH = -4       /* hipGetLastError() != hipSuccess after enqueue    */
};

int banet_version(void);
const char* banet_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * (1) EquationConstruction  -- replaces the TF op registered at utils.cu:150-171 and its
 *     kernel utils.cu:219-417 (5 cuBLAS sgemmBatched + 2 column reductions).
 *       jacobian   J [B,N,2,P]     gradient G [B,N,C,2]     difference d [B,N,C,1]
 *       left  AtA [B,P,P] = sum_n J_n^T (G_n^T G_n) J_n
 *       right Atb [B,P,1] = sum_n J_n^T  G_n^T d_n
 *     Never materialises the per-pixel PxP products (utils.cu:356-365 does: 22 GB/item at
 *     640x480, P=134).  Supported: 1 <= P <= 304, any C >= 1, any N >= 1 (all on the matrix-pipe kernels; 272 < P <= 304, cfg-5's P = 298: the 17-block pass + three jobs, round 5).
 * ------------------------------------------------------------------------------------- */
size_t banet_equation_construction_workspace_bytes(int B, int N, int C, int P);
int banet_equation_construction_f32(const float* jacobian, const float* gradient,
                                    const float* difference, float* left, float* right,
                                    int B, int N, int C, int P, void* ws, size_t ws_bytes,
                                    banet_stream_t stream);

/* (2) EquationConstructionGrad -- replaces utils.cu:420-428 (op) / :465-694 (kernel),
 *     bound as the op's gradient at bundlenet.py:79-82.
 *       left_grad g0 [B,P,P], right_grad g1 [B,P,1]
 *       A_n = G_n J_n ; dA_n = 2 A_n g0 + d_n g1^T   (alpha = 2.0 as utils.cu:651)
 *       jacobian_grad = G^T dA [B,N,2,P]; gradient_grad = dA J^T [B,N,C,2];
 *       difference_grad = A g1 [B,N,C,1]
 *     Workspace: optional.  With ws_bytes >= banet_equation_construction_grad_workspace_bytes (P <= 304, 256-byte
 *     aligned) the matrix-pipe kernels run; with ws = NULL (or for shapes whose size is 0) the first-generation
 *     kernel, same results to rounding.                                                   */
size_t banet_equation_construction_grad_workspace_bytes(int B, int N, int C, int P);
int banet_equation_construction_grad_f32(const float* jacobian, const float* gradient,
                                         const float* difference, const float* left_grad,
                                         const float* right_grad, float* jacobian_grad,
                                         float* gradient_grad, float* difference_grad, int B,
                                         int N, int C, int P, void* ws, size_t ws_bytes,
                                         banet_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused path.  One "level problem" = B independent windows at one pyramid level.
 * ------------------------------------------------------------------------------------- */
enum { /* banet_level_t.variant: which reference iteration is restated */
  BANET_LEGACY_LM = 0,     /* legacy/ba.py:226-345 CameraIteration2 (MLP, exponent 1+y,
                              N/sum(mask) scaling, QR, V, accept/reject)                  */
  BANET_LEGACY_FIXED = 1,  /* legacy/ba.py:148-214 CameraIteration (lambda=|avg|^2, no V)  */
  BANET_BUNDLE_CAMERA = 2, /* bundlenet.py:122-191 CameraIteration (pose only)            */
  BANET_BUNDLE = 3         /* bundlenet.py:193-278 BundleIteration (pose + depth basis)   */
};

/* banet_level_t.flags -- named overrides of the kernel / arithmetic-form selection (0 in production).  The BANET_FLAG_* bits are the
 * documented ones; every other bit is development A/B plumbing (tools/, profiles/) and must be 0 in a caller's code.
 *   BANET_FLAG_FORCE_PATCH_GATHER / _STRIP_GATHER / _QUAD_GATHER, BANET_FLAG_NO_QUAD_GATHER : run (never run) that C = 128 gather kernel
 *       at any launch size (parity tests compare the kernels at oracle sizes; banet_gather_selection reports what a level runs);
 *   BANET_FLAG_SYRK_F16 / BANET_FLAG_NO_SYRK_F16 : the arithmetic FORM of the depth-block contraction.  Default (neither bit): the LM loop
 *       (banet_lm_level_f32) runs the fp16 two-piece form -- fp32 operands as two scaled fp16 pieces, three products, ~2^-21 per
 *       product, <= 2e-7 per entry of the matrix's own scale; an absolute 2^-25 of the column bound for entries more than 17 octaves
 *       below it -- on throughput-bound launches (N * B >= 32 * 76 800 pixels; banet_level_t.policy decides whether B counts), and the
 *       exact form (three bf16 pieces, six products, fp32-exact) elsewhere; the single pass banet_ba_assemble_f32 always runs the
 *       exact form.  _SYRK_F16: the fp16 form at any launch size and in the single pass too; _NO_SYRK_F16: the exact form everywhere.
 *       banet_syrk_selection reports the form;
 *   BANET_FLAG_SYRK_THREE_PRODUCTS : OPT-IN, reduced precision -- the K = 128 depth-block contraction with the three largest
 *       bf16 products only (~2^-16 per product instead of fp32-exact).  Never the default, never what bench.py's `value`
 *       is measured with.
 * (Until round 4 this field was called `reserved_` and the bits BANET_DEV_*: same offset, same values.  The BANET_DEV_* ENUM names stay
 * as aliases; the STRUCT FIELDS were renamed -- `reserved_` -> `flags`, `pad_` -> `policy` -- which is source-breaking for a C / C++
 * caller that named them (binary layout unchanged; the Python ctypes mirror keeps a `reserved_` property).) */
enum {
  BANET_FLAG_FORCE_PATCH_GATHER = 1 << 9,
  BANET_FLAG_FORCE_STRIP_GATHER = 1 << 18,
  BANET_FLAG_FORCE_QUAD_GATHER = 1 << 25,   /* the 4x4-pixel-item gather of latency-bound launches, at any size */
  BANET_FLAG_NO_QUAD_GATHER = 1 << 30,      /* ... never (the tile kernels instead) */
  BANET_FLAG_SYRK_THREE_PRODUCTS = 1 << 29,
  BANET_FLAG_SYRK_F16 = 1 << 24,             /* the fp16 two-piece SYRK also in banet_ba_assemble_f32 and at any launch size */
  BANET_FLAG_NO_SYRK_F16 = (int)0x80000000,  /* ... never: the exact bf16 form everywhere */
  BANET_DEV_FORCE_PATCH_GATHER = BANET_FLAG_FORCE_PATCH_GATHER,
  BANET_DEV_FORCE_STRIP_GATHER = BANET_FLAG_FORCE_STRIP_GATHER,
  BANET_DEV_FORCE_QUAD_GATHER = BANET_FLAG_FORCE_QUAD_GATHER,
  BANET_DEV_NO_QUAD_GATHER = BANET_FLAG_NO_QUAD_GATHER,
  BANET_DEV_SYRK_THREE_PRODUCTS = BANET_FLAG_SYRK_THREE_PRODUCTS,
  BANET_DEV_SYRK_F16 = BANET_FLAG_SYRK_F16,
  BANET_DEV_NO_SYRK_F16 = BANET_FLAG_NO_SYRK_F16
};

/* banet_level_t.policy -- what the kernel / form selection may depend on.
 *   BANET_POLICY_THROUGHPUT (0, default): kernels, the SYRK form and the number of partial rows are chosen from the whole LAUNCH
 *       (N, K, pairs AND the batch B): fastest, and results are within rounding (<= 1e-5 relative on a solve, tested) of any other
 *       batching of the same windows -- but not bit-identical to them: a window solved in a batch of 8 and in a batch of 32 may run
 *       different gather kernels, SYRK forms and summation splits.
 *   BANET_POLICY_BATCH_INVARIANT (1): every such decision is taken from the LEVEL alone (N, K, pairs, as if B = BANET_CANONICAL_BATCH),
 *       so a window's bits do not depend on how many windows share its launch or on how a batch is sharded over GPUs (tested:
 *       batches of 1 / 8 / 32 bit-identical).  Small batches then run the kernels tuned for 32 windows: slower below ~8 windows.
 *       The reference has no such dependence either way (utils.cu:181-198 reduces per item).                                        */
enum { BANET_POLICY_THROUGHPUT = 0, BANET_POLICY_BATCH_INVARIANT = 1 };
#define BANET_CANONICAL_BATCH 32

typedef struct banet_level {
  int32_t B;            /* windows                                                        */
  int32_t N;            /* points per window                                              */
  int32_t C;            /* feature channels (C <= 256)                                    */
  int32_t K;            /* depth-basis coefficients; 0 for the pose-only variants; <= 256
                           (P = 6 pairs + K > ~190: the solve keeps its matrix in a workspace:
                           banet_lm_level_f32's, or banet_ba_solve_update_ws_f32's;
                           banet_ba_solve_update_f32 alone then returns BANET_ERR_UNSUPPORTED) */
  int32_t H, W;         /* target map height / width at this level                        */
  int32_t variant;      /* BANET_LEGACY_LM ... BANET_BUNDLE                               */
  int32_t dense;        /* 1: the N = H*W points are this level's own pixel grid          */
  int32_t tgt_has_grad; /* 1: target map is the reference's [f|gx|gy] 3C layout
                              (legacy/ba.py:116-118); 0: C channels, gradient computed in
                              the kernel from the map (same arithmetic as grad_fixed)      */
  int32_t normalize_rays; /* dense only: 1 = bundlenet.py:119, 0 = legacy/ba.py:33-34     */
  float scale;          /* dense only: level scale s (points, intrinsics = full-res / s)  */
  int32_t pairs;        /* target frames per window (F - 1); 0 or 1 = the reference's 2-frame
                           case.  pairs > 1 (bundle variants only) is the multi-frame window
                           of SURVEY.md 8(d): the key frame carries depth / basis / Wc, every
                           other frame its own pose; P = 6 pairs + K, parameter order
                           [pose_1 .. pose_pairs, depth]                                     */
  int32_t flags;        /* 0 in production; BANET_FLAG_* overrides of the kernel / arithmetic-form selection
                           (above); was `reserved_` until round 4 (same offset)                */
  int32_t policy;       /* BANET_POLICY_THROUGHPUT (0) or BANET_POLICY_BATCH_INVARIANT (1); was `pad_`,
                           which had to be 0: existing callers get the default                  */
  const float* src;     /* dense: source map [B,H,W,C];  sparse: conv1 [B,N,C]            */
  const float* tgt;     /* target maps [B,pairs,H,W,C] or [B,pairs,H,W,3C]                */
  const float* depth;   /* D  [B,N]   (legacy: z-depth; bundle: range along the ray)      */
  const float* basis;   /* Bs [B,N,K] or NULL when K == 0                                 */
  const float* rays;    /* sparse: p [B,3,N] (bundlenet.py:115-119); dense: NULL          */
  const float* fx;      /* sparse: per-point level intrinsics [B,N] each                  */
  const float* fy;
  const float* ox;
  const float* oy;
  const float* intr;    /* dense: full-resolution (fx,fy,ox,oy) per window [B,4]          */
} banet_level_t;

/* Five k=1 conv layers of the lambda predictor (bundlenet.py:168-172): filters [Cin,Cout]
 * row-major (the reference's [1,Cin,Cout] variable), biases [Cout]; C->2C->4C->2C->C->1. */
typedef struct banet_mlp {
  const float* w[5];
  const float* b[5];
} banet_mlp_t;

/* Per-window LM state carried across iterations and levels (device memory, caller-owned).
 *   R [B,pairs,9]  T [B,pairs,3]  Wc [B,K]   current estimate (input and output)
 *   iters [B] int32                 iterations executed at the current level
 *   ratio [B]                       legacy "keep ratio" (legacy/ba.py:214 / :344)
 *   lambda_out [B]                  last lambda (diagnostic)
 *   delta [B,P]                     last solved update (diagnostic / parity tests)       */
typedef struct banet_state {
  float* R;
  float* T;
  float* Wc;
  int32_t* iters;
  float* ratio;
  float* lambda_out;
  float* delta;
} banet_state_t;

/* (3) one fused assembly pass: warp -> sample -> residual/gradient -> Jacobians -> normal
 *     equations, for all B windows at the pose (R,T,Wc) -- replaces the TF graph of
 *     bundlenet.py:206-263 / legacy/ba.py:238-283 plus the EquationConstruction op.
 *       AtA [B,P,P], Atb [B,P]  (P = 6 pairs + K), absres [B,C] = sum over pairs and n of
 *       |d_nc|, nvalid [B] = sum over pairs of the in-image point counts                 */
size_t banet_ba_assemble_workspace_bytes(const banet_level_t* lv);
int banet_ba_assemble_f32(const banet_level_t* lv, const float* R, const float* T,
                          const float* Wc, float* AtA, float* Atb, float* absres,
                          float* nvalid, void* ws, size_t ws_bytes, banet_stream_t stream);
/* The same pass, additionally writing the in-image mask bit (bundlenet.py:155,231 / utils_python.py:61-117) the kernel decided
 * for every pixel: mask_out [B * pairs][N] bytes, 1 = in the image.  A parity diagnostic (bench.py's sweep gate compares it
 * bit by bit with the float64 oracle's mask, so that a float32-vs-float64 disagreement on a pixel sitting on the image border is
 * demonstrated, not assumed); same kernels, same sums.                                                                        */
int banet_ba_assemble_mask_f32(const banet_level_t* lv, const float* R, const float* T,
                               const float* Wc, float* AtA, float* Atb, float* absres, float* nvalid,
                               unsigned char* mask_out, void* ws, size_t ws_bytes, banet_stream_t stream);

/* (4) lambda prediction + damping + solve + SE(3)/W update for all B windows
 *     (bundlenet.py:165-190,241-276; legacy/ba.py:187-213,266-302).  Consumes the outputs
 *     of (3); updates `st` in place.  l2_base: bundlenet.py:252-253 (pass 1.0 for none). */
int banet_ba_solve_update_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base,
                              const float* AtA, const float* Atb, const float* absres,
                              const float* nvalid, banet_state_t* st, banet_stream_t stream);
/*     The same with a caller workspace for the systems whose matrix does not fit the LDS
 *     (P > ~190, e.g. K = 256): workspace_bytes = 0 means (4) alone is enough (ws may be NULL). */
size_t banet_ba_solve_update_workspace_bytes(const banet_level_t* lv);
int banet_ba_solve_update_ws_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base,
                                 const float* AtA, const float* Atb, const float* absres,
                                 const float* nvalid, banet_state_t* st, void* ws, size_t ws_bytes,
                                 banet_stream_t stream);

/* (5) the LM loop at one level, entirely enqueued (no host sync): max_iters iterations of
 *     (3)+(4).  For BANET_LEGACY_LM with early_termination != 0 the accept/reject test and
 *     the update-norm thresholds of legacy/ba.py:132-140,343-345 are evaluated on the
 *     device per window (the CheckUpdate pass of iteration k is the assembly pass of
 *     iteration k+1); st->iters receives the reference's iteration count.               */
size_t banet_lm_level_workspace_bytes(const banet_level_t* lv);
int banet_lm_level_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base,
                       int max_iters, int early_termination, banet_state_t* st, void* ws,
                       size_t ws_bytes, banet_stream_t stream);

/* (5b) the same loop with the reference's run-time LM configuration.  legacy/ba.py:5-9 keeps
 *     early_termination / angle_change / translation_change / residual_ratio / qr as module
 *     globals that its drivers overwrite (legacy/example.py:8, legacy/eval.py:9); here they are
 *     arguments.  banet_lm_level_f32 == banet_lm_level_ex_f32 with params = NULL (the
 *     reference's defaults, banet_lm_params_default).
 *       angle_change, translation_change : loop continues while both update norms exceed them
 *                                          (legacy/ba.py:133); only read with early_termination
 *       residual_ratio                   : accept iff avg' < residual_ratio * avg (ba.py:343)
 *       solver (legacy variants only)    : BANET_SOLVER_QR  = tf.qr + triangular solve
 *                                          (ba.py:205-206,292-293, `qr = True`);
 *                                          BANET_SOLVER_INVERSE = tf.matrix_inverse (LU with
 *                                          partial pivoting) then a product (ba.py:203,290,
 *                                          `qr = False`).  The bundlenet variants always use
 *                                          tf.matrix_solve's algorithm class (bundlenet.py:183,
 *                                          267) and ignore this field.                        */
enum { BANET_SOLVER_QR = 0, BANET_SOLVER_INVERSE = 1 };
typedef struct banet_lm_params {
  float angle_change;       /* legacy/ba.py:6  default 0.002 * (3.14 / 180)                   */
  float translation_change; /* legacy/ba.py:7  default 0.0002                                 */
  float residual_ratio;     /* legacy/ba.py:8  default 1.0                                    */
  int32_t solver;           /* legacy/ba.py:9  default BANET_SOLVER_QR                        */
} banet_lm_params_t;
void banet_lm_params_default(banet_lm_params_t* params);
int banet_lm_level_ex_f32(const banet_level_t* lv, const banet_mlp_t* mlp, float l2_base,
                          int max_iters, int early_termination, const banet_lm_params_t* params,
                          banet_state_t* st, void* ws, size_t ws_bytes, banet_stream_t stream);

/* (6) per-level preparation -- the step immediately before the LM loop.
 *   banet_resample_f32   data [B,H,W,C], warp [B,N,2] (x,y) -> out [B,N,C]
 *       mode BANET_RESAMPLE_ZERO_PAD : tf.contrib.resampler.resampler as called at
 *            bundlenet.py:290,320,343-344,385 (bilinear, taps outside the image contribute 0,
 *            points with x <= -1, y <= -1, x >= W or y >= H give 0)
 *       mode BANET_RESAMPLE_CLAMP    : interpolate2d2, legacy/utils_python.py:177-232
 *            (legacy/ba.py:115): weights from the unclamped floor, indices clamped
 *   banet_target_map_f32 img [B,H,W,C] -> [B,H,W,3C] = [f | gx | gy], grad_fixed with REFLECT
 *            padding (bundlenet.py:92-100,323-324; legacy/ba.py:17-25,116-118)
 *   banet_depth_output_f32 out [B,N] = init_depth [B,N] + basis [B,N,K] . Wc [B,K]
 *            (bundlenet.py:397)                                                            */
enum { BANET_RESAMPLE_ZERO_PAD = 0, BANET_RESAMPLE_CLAMP = 1 };
int banet_resample_f32(const float* data, const float* warp, float* out, int B, int N, int C,
                       int H, int W, int mode, banet_stream_t stream);
int banet_target_map_f32(const float* img, float* out, int B, int H, int W, int C,
                         banet_stream_t stream);
int banet_depth_output_f32(const float* init_depth, const float* basis, const float* Wc,
                           float* out, int B, int N, int K, banet_stream_t stream);

/* (7) differentiable layer support -- the C-wide part of one BundleIteration / CameraIteration in the reference's
 *     own tensor layout (bundlenet.py:230-243) and its adjoint (what TF autodiff derives from the same statements),
 *     so that a training graph never materialises samp [B,N,3C], diff, grad or J [B,N,2,P]:
 *   banet_sample_stats_f32      conv1 [B,N,C], conv2 [B,H,W,3C] = [f|gx|gy], px, py [B,N] ->
 *       stats [B,N,8] = (M11, M12, M22, g1, g2, mask, 0, 0) with M = G^T G, g = G^T d,
 *       d = (conv1 - samp_f) mask, G = [samp_gx, samp_gy] mask, mask = px in [0,W-1] and py in [0,H-1];
 *       absd_part [B, banet_sample_stats_blocks(N), C] = per-block sums of |d| (add them in order for sum_n |d|)
 *   banet_sample_stats_grad_f32 dstats [B,N,8] (first five used), dabs [B,C] = dL/d(sum_n |d|) ->
 *       dconv1 [B,N,C], dconv2 [B,H,W,3C] (ACCUMULATED with float atomics: zero it first), dpos [B,N,2] = dL/d(px,py)
 *     C <= 256, B H W 3C < 2^32.                                                                                   */
int banet_sample_stats_blocks(int N);
int banet_sample_stats_f32(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C,
                           int H, int W, float* stats, float* absd_part, banet_stream_t stream);
int banet_sample_stats_grad_f32(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N,
                                int C, int H, int W, const float* dstats, const float* dabs, float* dconv1,
                                float* dconv2, float* dpos, banet_stream_t stream);

/*   banet_sample_stats_grad_det_f32: the same gradients with dconv2 accumulated WITHOUT float atomics -- every point
 *     writes its 3C contribution row and (target cell, bilinear fractions); per texel the contributions are then gathered
 *     in a fixed order (cells row-major, ascending point index): bit-reproducible training gradients.  Workspace from
 *     banet_sample_stats_grad_workspace_bytes (3C + 12 floats per point, 16 bytes per texel).                            */
size_t banet_sample_stats_grad_workspace_bytes(int B, int N, int C, int H, int W);
int banet_sample_stats_grad_det_f32(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N,
                                    int C, int H, int W, const float* dstats, const float* dabs, float* dconv1,
                                    float* dconv2, float* dpos, void* ws, size_t ws_bytes, banet_stream_t stream);

/* (7b) backward of the fused dense assembly (3) -- SURVEY.md 8(f1).  Given the upstream gradients of ONE assembly pass at
 *     the state (R, T, Wc),
 *       gAtA [B,P,P] (any matrix: symmetrised internally, utils.cu:648-657 assumes symmetry), gAtb [B,P],
 *       gabs [B,C] = dL/d(sum_n |d_nc|)  (= dL/d avg / N for bundlenet.py:243),
 *     it ACCUMULATES (+=, so that the iterations of a level share one buffer; zero them first)
 *       dsrc [B,N,C], dmap3 [B,H,W,3C] = adjoint of the target's [f|gx|gy] map (bundlenet.py:323-324), ddepth [B,N],
 *       dbasis [B,N,K]
 *     and WRITES dpose [B, 12 + K] = (dL/dR [9], dL/dT [3], dL/dWc [K]) through the assembly (the caller adds the
 *     direct dependence of the update step on R, T, Wc).  What TF autodiff + EquationConstructionGrad (bundlenet.py:79-82,
 *     utils.cu:465-694) compute for bundlenet.py:206-263, per pixel, without J / G / d in memory.  Bit-reproducible:
 *     the target-map adjoint is gathered per texel in a fixed order (integer atomics only build the cell lists).
 *     Supported: dense = 1 with tgt_has_grad = 0 -- or (round 5) the reference's own sparse layout, dense = 0 with tgt_has_grad = 1:
 *     src = conv1 [B,N,C] at N sampled points, rays / fx / fy / ox / oy per point, tgt = the [f|gx|gy] map [B,H,W,3C]
 *     (bundlenet.py:332-399, what the reference trains on); dmap3 is then the gradient of that map itself --, pairs <= 1,
 *     C <= 256 (sparse: not C > 128 together with K > 128), and BANET_BUNDLE with 1 <= K <= 256 or the pose-only
 *     BANET_BUNDLE_CAMERA (bundlenet.py:122-191: K = 0, P = 6; basis / Wc / dbasis are not touched and may be NULL); else
 *     workspace_bytes = 0 and BANET_ERR_UNSUPPORTED.  A multi-frame window (pairs > 1) is the sum of its frames' two-frame
 *     terms: call once per target frame with the frame's sub-blocks of gAtA / gAtb (banet_amd/dense_train.py does).
 *   banet_target_map_adjoint_f32: dimg [B,H,W,C] += dmap3_f + grad_fixed^T (dmap3_gx, dmap3_gy) -- the adjoint of
 *     banet_target_map_f32 (REFLECT rim: zero gradient on the 1-px border, bundlenet.py:92-100); once per level.      */
size_t banet_dense_adjoint_workspace_bytes(const banet_level_t* lv);
/*   banet_spd_solve_f32: x [B,P] = A^-1 rhs for B symmetric positive definite systems A [B,P,P] (the blocked LDL^T of (4) as an op
 *     of its own; the matrix must fit the LDS: 32 <= P <= ~190, else BANET_ERR_UNSUPPORTED).  The backward of the layer calls it
 *     twice per iteration -- the damped system of bundlenet.py:264-267 again, then lam = A^-T g for the implicit-function gradient
 *     of tf.matrix_solve (dA = -lam x^T, db = lam).                                                                      */
int banet_spd_solve_f32(const float* A, const float* rhs, float* x, int B, int P, banet_stream_t stream);
int banet_dense_adjoint_f32(const banet_level_t* lv, const float* R, const float* T, const float* Wc, const float* gAtA,
                            const float* gAtb, const float* gabs, float* dsrc, float* dmap3, float* ddepth,
                            float* dbasis, float* dpose, void* ws, size_t ws_bytes, banet_stream_t stream);
/*   banet_dense_adjoint_ex_f32: the same with flags.  BANET_ADJOINT_OVERWRITE: dsrc / ddepth / dbasis are WRITTEN -- every
 *     entry, zeros where nothing contributes -- instead of accumulated; BANET_ADJOINT_OVERWRITE_MAP: the same for dmap3.  The first
 *     call of a level (its last iteration; per target frame for dmap3, the first frame only for the buffers the frames share)
 *     then needs no zero-filled buffers and reads none (25 GB of fills + 25 GB of reads per 32-window 640x480 level).  Same bits
 *     as accumulating into zeros.  banet_target_map_adjoint_ex_f32 with BANET_ADJOINT_OVERWRITE: dimg is written, not added to. */
/*   BANET_ADJOINT_FOLD_TARGET (round 6; dense layout only -- dense = 1, tgt_has_grad = 0): `dmap3` is the gradient of the TARGET MAP
 *     itself, [B,H,W,C] -- the adjoint of the bilinear sampling AND of grad_fixed (bundlenet.py:92-100, REFLECT rim -> 0) in one
 *     step, accumulated (BANET_ADJOINT_OVERWRITE_MAP: written, every texel) per small texel tile in LDS by the one wave that owns the
 *     tile, pixels in a fixed order (cells row-major, ascending pixel index): still no float atomics between waves, still
 *     bit-reproducible.  Neither the per-pixel 3C adjoint rows nor the [f|gx|gy] map adjoint exist, and banet_target_map_adjoint_f32
 *     is not called: 3C (N + HW) floats less memory per window, about a third of the backward's traffic.  The workspace layout
 *     differs: size it with banet_dense_adjoint_workspace_bytes_ex(lv, flags) (0 = unsupported: a window's H W C must stay below 2^30,
 *     the tile kernels use 32-bit byte offsets inside a window).  BANET_ADJOINT_TILE_SHAPE(k), k = 1 .. 13: development
 *     switch (A/B) -- the tile kernel and its tile shape (csrc/adjoint.hip::launch_adj_tile); 0 = the default.                                              */
/*   BANET_ADJOINT_REUSE_DEPTH_SEED (multi-frame windows: the calls for target frames 2 .. of one iteration): the caller states that
 *     the depth block of gAtA, the depth part of gAtb, lv->basis and the workspace are those of the PREVIOUS call -- then z2 = 2 S_dd b,
 *     zeta = b.S_dd b and e = gAtb_d.b of that call are still in the workspace and only the frame's q = S_cd b is computed (one
 *     column block of the GEMM-shaped piece instead of K/16 + 1).  Same bits as without the flag; ignored where it does not apply.   */
enum { BANET_ADJOINT_OVERWRITE = 1, BANET_ADJOINT_OVERWRITE_MAP = 2, BANET_ADJOINT_FOLD_TARGET = 4, BANET_ADJOINT_REUSE_DEPTH_SEED = 8 };
#define BANET_ADJOINT_TILE_SHAPE(k) (((k) & 15) << 4)
size_t banet_dense_adjoint_workspace_bytes_ex(const banet_level_t* lv, int flags);
int banet_dense_adjoint_ex_f32(const banet_level_t* lv, const float* R, const float* T, const float* Wc, const float* gAtA,
                               const float* gAtb, const float* gabs, float* dsrc, float* dmap3, float* ddepth,
                               float* dbasis, float* dpose, int flags, void* ws, size_t ws_bytes, banet_stream_t stream);
int banet_target_map_adjoint_f32(const float* dmap3, float* dimg, int B, int H, int W, int C, banet_stream_t stream);
int banet_target_map_adjoint_ex_f32(const float* dmap3, float* dimg, int B, int H, int W, int C, int flags,
                                    banet_stream_t stream);

/* (7c) backward of the SMALL part of one BundleIteration / CameraIteration (round 6) -- what tf.gradients derives for
 *     bundlenet.py:165-190 / :241-276 after the EquationConstruction op: avg = sum|d| / (N pairs) -> lambda MLP -> lambda = l2
 *     ||avg||^(2 + y) -> damping (bundle: last coefficient undamped, :264-266; bundle_camera: all six, no l2, :181-182) ->
 *     tf.matrix_solve -> SE(3) / W update (AngleaAxisRotation :17-37, VMatrix :39-46).  Four launches instead of a ~150-launch
 *     framework graph.  Inputs: the assembly outputs AtA [B,P,P], Atb [B,P], absres [B,C] (= sum |d|), the forward's solution
 *     delta [B,P] (banet_state_t.delta), the state BEFORE the update R [B,pairs,9], T [B,pairs,3], and the upstream gradients of
 *     the updated state gR [B,pairs,9], gT [B,pairs,3], gW [B,K].  Outputs (written): gAtA [B,P,P] (not symmetrised -- (7b) does
 *     that), gAtb [B,P], gabs [B,C] = dL/d absres, dR [B,pairs,9], dT [B,pairs,3] = the direct dependence of (R', T') on (R, T);
 *     dL/dWc = gW (W' = W + sol) is the caller's.  gmlp: the ten lambda-weight gradients, ACCUMULATED (+=, summed over the
 *     windows in window order: bit-reproducible) -- zero them before the first iteration of a level.  The damped system is solved
 *     by implicit differentiation on banet_spd_solve_f32's kernel (lam = A^-1 dL/dsol, dL/dA = -lam sol^T, dL/dAtb = lam): P must
 *     fit it (32 <= P <= ~190) or be < 32 (pose only: in-kernel Cholesky); C <= 256; variant BANET_BUNDLE (K >= 1) or
 *     BANET_BUNDLE_CAMERA (K = 0); else workspace_bytes = 0 and BANET_ERR_UNSUPPORTED.                                        */
size_t banet_small_step_adjoint_workspace_bytes(int variant, int B, int N, int C, int K, int pairs);
int banet_small_step_adjoint_f32(int variant, int B, int N, int C, int K, int pairs, float l2_regularizer_base,
                                 const banet_mlp_t* mlp, const float* AtA, const float* Atb, const float* absres,
                                 const float* delta, const float* R, const float* T, const float* gR, const float* gT,
                                 const float* gW, float* gAtA, float* gAtb, float* gabs, float* dR, float* dT,
                                 const banet_mlp_t* gmlp, void* ws, size_t ws_bytes, banet_stream_t stream);

/* (8) optional kernel timing, used by bench.py for the roofline figure.  Between
 *     banet_profile_begin and banet_profile_end every launch of the fused assembly kernel
 *     (from banet_ba_assemble_f32 / banet_lm_level_f32) is bracketed by two hipEvents recorded
 *     on its stream.  banet_profile_end synchronises those events and returns, per distinct
 *     level size (tag = points per window N), the number of launches and the summed kernel
 *     time in milliseconds.  Process-global and NOT thread-safe: a measurement aid, not part
 *     of the data path; events are created in _begin, never inside the timed region.       */
int banet_profile_begin(int max_launches);
int banet_profile_end(int max_tags, int32_t* tag_points, int32_t* tag_launches, double* tag_ms,
                      int32_t* ntags);
/*   banet_build_id: a 16-hex-digit digest of the kernel sources + compile flags this library was built from (build.sh).
 *     Measurement provenance only: bench.py refuses a PMC traffic file (profiles/pmc_traffic.json) recorded on another build.
 *   banet_profile_ranges: 1 = wrap every kernel launch of the assembly / solve path in a roctx range ("banet.gather N=...",
 *     "banet.syrk", "banet.fold", "banet.reduce", "banet.solve") so that `rocprofv3 --marker-trace` groups the kernels by
 *     role (the reference has no tracing hooks at all).  Returns 0 when the roctx library could not be
 *     loaded.  Off by default; also enabled by the environment variable BANET_ROCTX=1.                                     */
const char* banet_build_id(void);
/*   banet_gather_selection: which assembly (gather) kernel a level of this shape runs -- 0 = ba_gather_kernel (generic),
 *     1 = ba_gather128_kernel (C = 128, 8x8 tiles, direct taps), 2 = ba_gather128p_kernel (wave-private LDS patches),
 *     3 = ba_gather128s_kernel (16x16 strip segments, rolling LDS window), 4 = ba_gather128q_kernel (4x4-pixel items, one step
 *     per item: latency-bound launches); negative = error code.  The selection depends on
 *     the batch (work items per resident wave), so bench.py / the tests use this to run a one-window parity check on the
 *     kernel the full batch runs (the BANET_FLAG_FORCE_* / _NO_QUAD_GATHER bits of banet_level_t.flags).                */
int banet_gather_selection(const banet_level_t* lv);
/*   banet_syrk_selection: which depth-block contraction (SYRK) kernel banet_lm_level_f32 runs for this level AND batch size --
 *     0 = ba_syrk_kernel (LDS-tiled fp32 MFMA), 1 = ba_syrk_direct_kernel (fp32 MFMA, A/B), 2 = ba_syrk_bf16x6_kernel (fp32 operands
 *     split exactly into three bf16 pieces, six products), 3 = the syrk_wide.hip jobs (K = 256 / many frames), 4 =
 *     ba_syrk_bf16x6_kernel's fp16 two-piece form (three products, per-column power-of-two scales: throughput-bound launches of
 *     the LM loop; BANET_DEV_SYRK_F16 runs it in a single assembly pass too, BANET_DEV_NO_SYRK_F16 never); -1000 = no depth block
 *     (K = 0); other negative = error code.                                                                                    */
int banet_syrk_selection(const banet_level_t* lv);
int banet_profile_ranges(int enable);

#ifdef __cplusplus
}
#endif
#endif /* BANET_HIP_H_ */
