// Backward of the fused dense bundle assembly: given the upstream gradients of one assembly pass,
//   gAtA [B,P,P], gAtb [B,P], gabs [B,C] = dL / d(AtA, Atb, sum_n |d|),
// the gradients with respect to everything the pass read -- source map, target map, depth, depth basis, pose (R, T)
// and depth coefficients Wc.  It is what TF autodiff + the registered EquationConstructionGrad (bundlenet.py:79-82,
// utils.cu:465-694) compute for the statements bundlenet.py:206-263, restated per pixel without J, G or d in memory
// (derivation and float64 statement: dense_adjoint.py of the test oracle, validated there against finite differences).
//
// With S = (gAtA + gAtA^T)/2, b = the pixel's basis row, J = [Jc | jd b^T], M = G^T G, g = G^T d:
//   q = S_cd b, z = S_dd b, zeta = b.z, e = gAtb_d.b, t = Jc q + jd zeta
//   dM = (Jc S_cc + jd q^T) Jc^T + t jd^T,  dg = Jc gAtb_c + jd e          -> channel adjoints of (gx, gy, d)
//   dJc = 2 M (Jc S_cc + jd q^T) + g gAtb_c^T,  djd = 2 M t + g e          -> geometry adjoint -> dD, dR, dT
//   dbasis = 2 s z + 2 S_cd^T u + r gAtb_d + dD Wc   (u, s, r = the forward's per-pixel records)
// Kernels, in launch order:
//   adj_sym_kernel      S = (gAtA + gAtA^T)/2
//   adj_basis_kernel    the one GEMM-shaped piece, [N x K] . [K x (K+7)] per window on v_mfma_f32_16x16x4_f32 (exact fp32):
//                       z2 = 2 S_dd b (written), per-pixel records (q, zeta, e)
//   adj_pixel_kernel    one wave per pixel, lane = channel / coefficient: recomputes depth, warp, taps, M, g; writes
//                       dsrc, ddepth, dbasis (+=: every pixel is owned by one wave), the pixel's 3C adjoint row of the sampled
//                       [f|gx|gy] vector, its target cell + bilinear fractions; per-wave partial sums of dR, dT, dWc
//   adj_scan / adj_fill target cell -> list of the source pixels whose bilinear footprint starts there (integer atomics only)
//   adj_map_kernel      one wave per target texel: GATHERS the (<= 4 cells) contributions in ascending pixel order into the
//                       [f|gx|gy] map adjoint -- no float atomics, bit-reproducible (the sparse-layout ba_sample_stats_grad_kernel
//                       scatters with atomics instead)
//   adj_fold_kernel     per-wave partials -> dpose [B, 12 + K] in fixed order
// and, once per level after all iterations, target_map_adjoint_kernel: d tgt += dmap_f + grad_fixed^T (dmap_gx, dmap_gy)
// (REFLECT rim: bundlenet.py:92-100).
#include <type_traits>
#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"
#include "syrk_split.hpp"

namespace banet {

namespace {

typedef __bf16 bf16x8a __attribute__((ext_vector_type(8)));
constexpr int kAdjMaxCJ = 4;    // C <= 256
constexpr int kAdjMaxKJ = 4;    // K <= 256 (K <= 128: adj_basis_kernel keeps the whole K x (K+16) seed block in LDS; above: column chunks)
constexpr int kAdjHdr = 16;     // partial row: dR (9), dT (3), pad, then dWc (K)
constexpr int kScanThreads = 1024;
constexpr int kFrac = 8;        // per-pixel record of the map adjoint: key (int; -1 = outside), ax, ay, dg1, dg2, dM11, dM12, dM22

struct AdjArgs {
  banet_level_t lv;
  const float* R;
  const float* T;
  const float* Wc;
  const float* S;     // [B][P][P]
  const float* gb;    // [B][P]
  const float* gabs;  // [B][C]
  float* z2;          // [B][N][K]
  float* arec;        // [B][N][8]   q0..q5, zeta, e
  float* arow;        // [B][N][3C]  (nullptr in fold mode: the rows are never materialised)
  float* frac;        // [B][N][kFrac]   key (int), ax, ay, dg1, dg2, dM11, dM12, dM22
  int* cnt;           // [B][HW]
  int* start;         // [B][HW][2]  (start, count)
  int* cursor;        // [B][HW]
  int* list;          // [B][N]
  float* part;        // [B][G * 4][kAdjHdr + K]
  int G;
  float* dsrc;
  float* ddepth;
  float* dbasis;
  float* dmap3;
  float* dpose;
  int overwrite;      // 1: dsrc / ddepth / dbasis are WRITTEN (every entry, zeros where nothing contributes) instead of accumulated
  int overwrite_map;  // 1: the same for dmap3
  // fold mode (round 6, adj_tile_kernel): dmap3 is the target map's gradient itself, [B][H][W][C]
  float* lrec;        // [B][N][kFrac]  the records in cell-list order, key replaced by the pixel index
  int* list2;         // [B][N]         scratch of the big-cell sort
  int2* lidx;         // [B][N]         (pixel index, x0 | y0 << 16) in cell-list order: what adj_tile2_kernel addresses with
  int* bigq;          // [1 + B*N/32]   queue of the cells with more than kSortSerial entries (bigq[0] = their number)
  int reuse_z;        // BANET_ADJOINT_REUSE_DEPTH_SEED: z2 / zeta / e of the previous call on this workspace are valid (adj_basis6_kernel)
};

__global__ void adj_sym_kernel(const float* __restrict__ g, float* __restrict__ S, int P, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const size_t b = e / ((size_t)P * P);
  const int r = (int)(e - b * P * P);
  const int i = r / P, j = r - i * P;
  S[e] = 0.5f * (g[e] + g[b * P * P + (size_t)j * P + i]);
}

// ---- the GEMM-shaped piece --------------------------------------------------------------------------------------------
template <int NK>   // KP = 16 NK >= K
__global__ __launch_bounds__(kBlock, 2) void adj_basis_kernel(const AdjArgs a) {
  extern __shared__ float Wl[];
  constexpr int KP = 16 * NK, LS = KP + 20, NB = NK + 1;   // LS: 4 kq groups 4 LS floats apart -> 16 banks apart
  const int b = blockIdx.y, K = a.lv.K, N = a.lv.N, P = 6 + K;
  {
    const float* __restrict__ S = a.S + (size_t)b * P * P;
    const float* __restrict__ gb = a.gb + (size_t)b * P;
    for (int idx = threadIdx.x; idx < KP * LS; idx += kBlock) {
      const int k = idx / LS, j = idx - k * LS;
      float v = 0.f;
      if (k < K) {
        if (j < K)
          v = S[(size_t)(6 + k) * P + 6 + j];
        else if (j >= KP && j < KP + 6)
          v = S[(size_t)(j - KP) * P + 6 + k];
        else if (j == KP + 6)
          v = gb[6 + k];
      }
      Wl[idx] = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = wave_id(), i = lane & 15, kq = lane >> 4;
  const float* __restrict__ bas = a.lv.basis + (size_t)b * N * K;
  float* __restrict__ z2 = a.z2 + (size_t)b * N * K;
  float* __restrict__ arec = a.arec + (size_t)b * N * 8;
  const int nrb = (N + 15) >> 4;
  for (int rb = blockIdx.x * kNumWaves + w; rb < nrb; rb += gridDim.x * kNumWaves) {
    const int n = rb * 16 + i;
    const bool okn = n < N;
    const float* __restrict__ row = bas + (size_t)(okn ? n : 0) * K;
    float av[NK][4];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 16 * kk + 4 * kq + e;
        const float v = row[k < K ? k : 0];
        av[kk][e] = (okn && k < K) ? v : 0.f;
      }
    f32x4 acc[NB];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) acc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      asm volatile("" ::: "memory");   // keeps the LDS operand reads inside the row-block loop (LICM would hoist all 36 NK of them)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wl = Wl + (16 * kk + 4 * kq + e) * LS + i;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk][e], wl[16 * jb], acc[jb], 0, 0, 0);
      }
    }
    // accumulator layout: lane (j = lane & 15, rq = lane >> 4) holds rows 4 rq + v, column 16 jb + j
    float zeta[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int nn = rb * 16 + 4 * kq + v;
      const bool ok = nn < N;
#pragma unroll
      for (int jb = 0; jb < NK; ++jb) {
        const int j = 16 * jb + i;
        if (ok && j < K) {
          zeta[v] = fmaf(acc[jb][v], bas[(size_t)nn * K + j], zeta[v]);
          z2[(size_t)nn * K + j] = 2.f * acc[jb][v];
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float zt = row16_sum(zeta[v]);
      const int nn = rb * 16 + 4 * kq + v;
      if (nn < N) {
        if (i < 6) arec[(size_t)nn * 8 + i] = acc[NK][v];
        if (i == 6) arec[(size_t)nn * 8 + 7] = acc[NK][v];
        if (i == 7) arec[(size_t)nn * 8 + 6] = zt;
      }
    }
  }
}

// The same GEMM on the bf16 matrix pipe at fp32 accuracy (K <= 128): both operands are split exactly into three bf16 pieces and
// each 16 x 16 x 32 block takes six v_mfma_f32_16x16x32_bf16 (products exact in fp32, fp32 accumulate, the three terms below
// fp32 rounding dropped -- the scheme of syrk.hip / eqcon_grad.hip): 6 x 17 cycles against 8 x 32 for v_mfma_f32_16x16x4_f32.
// The seed block is split ONCE per workgroup into LDS as MFMA B operands ([k step][column block][piece][lane], 3 KB per block and
// k step: 108 KB at K = 128); a wave takes 16 pixels at a time: lane (m, kq) loads the 8 coefficients 32 ks + 8 kq .. + 7 of pixel
// m (32 contiguous bytes; the four kq lanes of a pixel read 128), splits them (registers) and accumulates.  512 threads: two
// waves per SIMD.  Epilogue as adj_basis_kernel (same accumulator layout).  flags bit 26 keeps the fp32-MFMA kernel (A/B).
constexpr int kAdjB6Threads = 512, kAdjB6Waves = kAdjB6Threads / 64;
// REUSE (BANET_ADJOINT_REUSE_DEPTH_SEED; the later target frames of a multi-frame window): S_dd, gAtb_d and the basis are those of the
// previous call on this workspace, so z2 = 2 S_dd b, zeta and e are already there -- only the frame's own q = S_cd b is computed:
// one column block of MFMAs instead of NK + 1, no z2 traffic.
template <int NK, bool REUSE>   // KP = 16 NK >= K, NK <= 8
__global__ __launch_bounds__(kAdjB6Threads) void adj_basis6_kernel(const AdjArgs a) {
  constexpr int KS = (NK + 1) / 2, NB = NK + 1, KP = 16 * NK, JB0 = REUSE ? NK : 0;      // column blocks JB0 .. NK
  extern __shared__ __attribute__((aligned(16))) unsigned sB6[];   // [KS][NB][3][64] quads
  const int b = blockIdx.y, K = a.lv.K, N = a.lv.N, P = 6 + K;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  {
    const float* __restrict__ S = a.S + (size_t)b * P * P;
    const float* __restrict__ gb = a.gb + (size_t)b * P;
    for (int task = w; task < KS * NB; task += kAdjB6Waves) {
      const int ks = task / NB, jb = task - ks * NB;
      if (jb < JB0) continue;
      float vv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * ks + 8 * kq + e, j = 16 * jb + m;     // seed[k][j], the layout of adj_basis_kernel's Wl
        float v = 0.f;
        if (k < K) {
          if (j < K)
            v = S[(size_t)(6 + k) * P + 6 + j];
          else if (j >= KP && j < KP + 6)
            v = S[(size_t)(j - KP) * P + 6 + k];
          else if (j == KP + 6)
            v = gb[6 + k];
        }
        vv[e] = v;
      }
      u32x4_t pc[3];
      split8_bf16x3(vv, pc);
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x4_t*>(&sB6[(((ks * NB + jb) * 3 + t) * 64 + lane) * 4]) = pc[t];
    }
  }
  __syncthreads();
  const float* __restrict__ bas = a.lv.basis + (size_t)b * N * K;
  float* __restrict__ z2 = a.z2 + (size_t)b * N * K;
  float* __restrict__ arec = a.arec + (size_t)b * N * 8;
  const int nrb = (N + 15) >> 4;
  const bool k8 = (K & 7) == 0;
  auto load_a = [&](int rb, int ks, float (&v)[8]) __attribute__((always_inline)) {
    const int n = min(rb * 16 + m, N - 1), c0 = 32 * ks + 8 * kq;
    const float* p = bas + (size_t)n * K + c0;
    if (k8 && c0 + 8 <= K) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p), t1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = t0[e];
        v[4 + e] = t1[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (c0 + e < K) ? p[e] : 0.f;
    }
    if (rb * 16 + m >= N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
  };
  for (int rb = blockIdx.x * kAdjB6Waves + w; rb < nrb; rb += gridDim.x * kAdjB6Waves) {
    f32x4 acc[NB];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) acc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av[8], an[8];
    load_a(rb, 0, av);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("" ::: "memory");   // the B operands are re-read from LDS every row block (hoisting all of them spills)
      u32x4_t pa[3];
      split8_bf16x3(av, pa);
      if (ks + 1 < KS) load_a(rb, ks + 1, an);
      constexpr int kTa[6] = {2, 0, 1, 1, 0, 0}, kTb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
      u32x4_t pb[NB][3];
#pragma unroll
      for (int jb = JB0; jb < NB; ++jb)
#pragma unroll
        for (int t = 0; t < 3; ++t) pb[jb][t] = *reinterpret_cast<const u32x4_t*>(&sB6[(((ks * NB + jb) * 3 + t) * 64 + lane) * 4]);
      // round 6: the SEED block is the A operand and the basis rows the B operand -- S_dd is symmetric and the extra block is kept
      // transposed, so the LDS image serves either way -- which transposes the product: lane (m, rq) then holds coefficients
      // 16 jb + 4 rq .. + 3 of pixel m, four CONSECUTIVE floats: z2 and the basis re-read of zeta are 16-byte accesses (the old layout
      // -- lane = column, four rows -- needed 32 scattered 4-byte loads and stores per block: 56 % of the kernel's time by ablation)
#pragma unroll
      for (int t = 0; t < 6; ++t)       // term-major: consecutive MFMAs write different accumulators
#pragma unroll
        for (int jb = JB0; jb < NB; ++jb)
          acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8a, pb[jb][kTb[t]]), __builtin_bit_cast(bf16x8a, pa[kTa[t]]),
                                                            acc[jb], 0, 0, 0);
      if (ks + 1 < KS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = an[e];
      }
    }
    // accumulator layout: lane (m = pixel, rq = kq) holds z[pixel m][16 jb + 4 rq + v], v = 0..3
    const int nn = rb * 16 + m;
    const bool ok = nn < N;
    const size_t ro = (size_t)(ok ? nn : 0) * K;
    float zeta = 0.f;
    if constexpr (REUSE) {
      if (ok) {                            // the frame's q; (zeta, e) stay as the first frame's call left them
        float* __restrict__ ar = arec + (size_t)nn * 8;
        if (kq == 0) *reinterpret_cast<f32x4*>(ar) = acc[NK];
        if (kq == 1) {
          ar[4] = acc[NK][0];
          ar[5] = acc[NK][1];
        }
      }
      continue;
    }
    if ((K & 3) == 0) {
#pragma unroll
      for (int jb = 0; jb < NK; ++jb) {
        const int j = 16 * jb + 4 * kq;
        if (ok && j < K) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bas + ro + j);
          zeta += acc[jb][0] * bq[0] + acc[jb][1] * bq[1] + acc[jb][2] * bq[2] + acc[jb][3] * bq[3];
          *reinterpret_cast<f32x4*>(z2 + ro + j) = 2.f * acc[jb];
        }
      }
    } else {
#pragma unroll
      for (int jb = 0; jb < NK; ++jb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int j = 16 * jb + 4 * kq + v;
          if (ok && j < K) {
            zeta = fmaf(acc[jb][v], bas[ro + j], zeta);
            z2[ro + j] = 2.f * acc[jb][v];
          }
        }
    }
    zeta += __shfl_xor(zeta, 16, 64);      // the pixel's four coefficient quarters (lanes m, m + 16, m + 32, m + 48)
    zeta += __shfl_xor(zeta, 32, 64);
    if (ok) {                              // the extra block: rows 0..5 = q = S_cd b, row 6 = e = gAtb_d . b  (row 4 rq + v)
      float* __restrict__ ar = arec + (size_t)nn * 8;
      if (kq == 0) *reinterpret_cast<f32x4*>(ar) = acc[NK];
      if (kq == 1) {
        ar[4] = acc[NK][0];
        ar[5] = acc[NK][1];
        ar[7] = acc[NK][2];
        ar[6] = zeta;
      }
    }
  }
}

// K > 128: the seed block [K x (K+16)] no longer fits the LDS (K = 256: 282 KB), so its COLUMN blocks are taken in chunks of NBC
// (K = 256: 3 chunks of 6 / 6 / 5 blocks, 116 KB each) and the row blocks are walked once per chunk -- the basis is read NCH
// times (the backward of a K = 256 level is not the hot path; the forward reads it twice per iteration anyway).  A wave meets the
// same row blocks in every chunk, so zeta = b.S_dd b is accumulated in the pixel's record by the lane that owns it: first chunk
// writes, later chunks add, in chunk order -- deterministic.
template <int NK, int NBC>
__global__ __launch_bounds__(kBlock, 1) void adj_basis_wide_kernel(const AdjArgs a) {
  extern __shared__ float Wl[];
  constexpr int KP = 16 * NK, LSC = 16 * NBC + 20, NB = NK + 1, NCH = (NB + NBC - 1) / NBC;   // LSC mod 32 = 20 as in adj_basis_kernel
  static_assert((NBC & 1) == 0, "bank spreading of the kq groups needs 16 NBC = 0 mod 32");
  const int b = blockIdx.y, K = a.lv.K, N = a.lv.N, P = 6 + K;
  const float* __restrict__ S = a.S + (size_t)b * P * P;
  const float* __restrict__ gb = a.gb + (size_t)b * P;
  const int lane = threadIdx.x & 63, w = wave_id(), i = lane & 15, kq = lane >> 4;
  const float* __restrict__ bas = a.lv.basis + (size_t)b * N * K;
  float* __restrict__ z2 = a.z2 + (size_t)b * N * K;
  float* __restrict__ arec = a.arec + (size_t)b * N * 8;
  const int nrb = (N + 15) >> 4;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    __syncthreads();   // the previous chunk's readers are done
    for (int idx = threadIdx.x; idx < KP * LSC; idx += kBlock) {
      const int k = idx / LSC, jj = idx - k * LSC, j = 16 * NBC * ch + jj;
      float v = 0.f;
      if (k < K && jj < 16 * NBC) {
        if (j < K)
          v = S[(size_t)(6 + k) * P + 6 + j];
        else if (j >= KP && j < KP + 6)
          v = S[(size_t)(j - KP) * P + 6 + k];
        else if (j == KP + 6)
          v = gb[6 + k];
      }
      Wl[idx] = v;
    }
    __syncthreads();
    for (int rb = blockIdx.x * kNumWaves + w; rb < nrb; rb += gridDim.x * kNumWaves) {
      const int n = rb * 16 + i;
      const bool okn = n < N;
      const float* __restrict__ row = bas + (size_t)(okn ? n : 0) * K;
      float av[NK][4];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 16 * kk + 4 * kq + e;
          const float v = row[k < K ? k : 0];
          av[kk][e] = (okn && k < K) ? v : 0.f;
        }
      f32x4 acc[NBC];
#pragma unroll
      for (int jb = 0; jb < NBC; ++jb) acc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        asm volatile("" ::: "memory");   // as in adj_basis_kernel: keep the LDS operand reads inside the loop
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wl = Wl + (16 * kk + 4 * kq + e) * LSC + i;
#pragma unroll
          for (int jb = 0; jb < NBC; ++jb)
            if (NBC * ch + jb < NB) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk][e], wl[16 * jb], acc[jb], 0, 0, 0);
        }
      }
      float zeta[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int nn = rb * 16 + 4 * kq + v;
        const bool ok = nn < N;
#pragma unroll
        for (int jb = 0; jb < NBC; ++jb) {
          const int j = 16 * (NBC * ch + jb) + i;
          if (NBC * ch + jb < NK && ok && j < K) {
            zeta[v] = fmaf(acc[jb][v], bas[(size_t)nn * K + j], zeta[v]);
            z2[(size_t)nn * K + j] = 2.f * acc[jb][v];
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float zt = row16_sum(zeta[v]);
        const int nn = rb * 16 + 4 * kq + v;
        if (nn < N) {
          if (i == 7) {
            if (ch == 0)
              arec[(size_t)nn * 8 + 6] = zt;
            else
              arec[(size_t)nn * 8 + 6] += zt;
          }
          if (NBC * ch <= NK && NK < NBC * (ch + 1)) {   // the chunk that holds the extra block [S_cd^T | gb]
            constexpr int je = NK % NBC;
            if (i < 6) arec[(size_t)nn * 8 + i] = acc[je][v];
            if (i == 6) arec[(size_t)nn * 8 + 7] = acc[je][v];
          }
        }
      }
    }
  }
}

// ---- per-pixel adjoint ------------------------------------------------------------------------------------------------
// sum over the wave, every lane gets it: DPP row sums + 4 v_readlane (the ds_bpermute butterfly of wave_sum costs ~6 LDS
// round trips per value; this kernel needs 8 sums per pixel).  Fixed order -> deterministic.
__device__ __forceinline__ float wsum(float v) {
  v = row16_sum(v);
  const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a0 + a1) + (a2 + a3);
}

// SP (round 5): SPARSE points in the reference's own layout -- what bundlenet.py:332-399 trains on: conv1 [B,N,C] sampled at N points,
// rays p [B,3,N] and per-point level intrinsics supplied (bundlenet.py:115-119), the target map precomputed as [f|gx|gy]
// [B,H,W,3C] (bundlenet.py:92-100) and sampled with plain bilinear taps, clamped one by one like the forward's generic kernel
// (gather.hip, GRAD = true).  Same per-pixel algebra; dmap3 is then the gradient of that 3C map itself.
template <int CJ, int KJ, bool SP = false>   // C <= 64 CJ, K <= 64 KJ (K > 128: 2 waves per SIMD -- the per-coefficient state would spill at 3)
__global__ __launch_bounds__(kBlock, (KJ >= 3 ? 2 : 3)) void adj_pixel_kernel(const AdjArgs a) {
  const banet_level_t& lv = a.lv;
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, w = wave_id();
  const int N = lv.N, C = lv.C, K = lv.K, H = lv.H, W = lv.W, P = 6 + K;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ tgt_b = lv.tgt + (size_t)b * H * W * (SP ? 3 * C : C);
  const float* __restrict__ bas_b = lv.basis + (size_t)b * N * K;
  const float* __restrict__ S = a.S + (size_t)b * P * P;
  const float* __restrict__ gb = a.gb + (size_t)b * P;
  float Scc[6][6], gbc[6], Rm[9], Tv[3];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    gbc[i] = gb[i];
#pragma unroll
    for (int j = 0; j < 6; ++j) Scc[i][j] = S[i * P + j];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) Rm[i] = a.R[b * 9 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) Tv[i] = a.T[b * 3 + i];
  float fx0 = 1.f, fy0 = 1.f, ox0 = 0.f, oy0 = 0.f;
  if constexpr (!SP) fx0 = lv.intr[b * 4 + 0], fy0 = lv.intr[b * 4 + 1], ox0 = lv.intr[b * 4 + 2], oy0 = lv.intr[b * 4 + 3];
  float fx = fx0 / lv.scale, fy = fy0 / lv.scale, ox = ox0 / lv.scale, oy = oy0 / lv.scale;     // SP: per point, below
  float wc[KJ], scd[KJ][6], gbd[KJ], dwc[KJ];
  bool kok[KJ], cok[CJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    const int k = lane + 64 * j;
    kok[j] = k < K;
    wc[j] = kok[j] ? a.Wc[(size_t)b * K + k] : 0.f;
    gbd[j] = kok[j] ? gb[6 + k] : 0.f;
    dwc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) scd[j][i] = kok[j] ? S[i * P + 6 + k] : 0.f;
  }
  float ga[CJ];
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int c = lane + 64 * j;
    cok[j] = c < C;
    ga[j] = cok[j] ? a.gabs[(size_t)b * C + c] : 0.f;
  }
  float accR[9], accT[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) accR[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) accT[i] = 0.f;

  const int nw = a.G * kNumWaves, chunk = (N + nw - 1) / nw;
  const int n_lo = (g * kNumWaves + w) * chunk, n_hi = min(N, n_lo + chunk);
  for (int n = n_lo; n < n_hi; ++n) {
    // ---- depth and warp (bundlenet.py:206-224), every lane the same values
    float p0, p1, p2;
    if constexpr (SP) {
      const size_t o = (size_t)b * 3 * N, qn = (size_t)b * N + n;
      p0 = lv.rays[o + n], p1 = lv.rays[o + N + n], p2 = lv.rays[o + 2 * (size_t)N + n];
      fx = lv.fx[qn], fy = lv.fy[qn], ox = lv.ox[qn], oy = lv.oy[qn];
    } else {
      const int qy = n / W, qx = n - qy * W;
      p0 = ((float)qx * lv.scale - ox0) / fx0, p1 = ((float)qy * lv.scale - oy0) / fy0, p2 = 1.f;
      if (lv.normalize_rays) {
        const float inv = 1.f / sqrtf(fmaxf(p0 * p0 + p1 * p1 + p2 * p2, 1e-12f));
        p0 *= inv;
        p1 *= inv;
        p2 *= inv;
      }
    }
    float bv[KJ], z2v[KJ], f1v[CJ], bsum = 0.f;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = kok[j] ? lane + 64 * j : 0;
      bv[j] = bas_b[(size_t)n * K + k];
      z2v[j] = a.z2[((size_t)b * N + n) * K + k];
      bv[j] = kok[j] ? bv[j] : 0.f;
      bsum = fmaf(bv[j], wc[j], bsum);
    }
#pragma unroll
    for (int j = 0; j < CJ; ++j) f1v[j] = src_b[(size_t)n * C + (cok[j] ? lane + 64 * j : 0)];
    const float* __restrict__ ar = a.arec + ((size_t)b * N + n) * 8;
    float q[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) q[i] = ar[i];
    const float zeta = ar[6], ee = ar[7];
    const float D = lv.depth[(size_t)b * N + n] + wsum(bsum);
    const float rx = Rm[0] * p0 + Rm[1] * p1 + Rm[2] * p2;
    const float ry = Rm[3] * p0 + Rm[4] * p1 + Rm[5] * p2;
    const float rz = Rm[6] * p0 + Rm[7] * p1 + Rm[8] * p2;
    const float X = rx * D + Tv[0], Y = ry * D + Tv[1], Z = rz * D + Tv[2];
    const float x = X / Z, y = Y / Z;
    const float px = fx * x + ox, py = fy * y + oy;
    const bool m = (px >= 0.f) && (px <= (float)(W - 1)) && (py >= 0.f) && (py <= (float)(H - 1));
    float* __restrict__ fr = a.frac + ((size_t)b * N + n) * kFrac;
    if (!m) {   // wave-uniform: no contribution to anything (M = g = 0)
      if (lane == 0) fr[0] = __int_as_float(-1);
      if (a.overwrite) {   // nothing was zero-filled: this pixel's rows are written here
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          if (cok[j]) a.dsrc[((size_t)b * N + n) * C + lane + 64 * j] = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j)
          if (kok[j]) a.dbasis[((size_t)b * N + n) * K + lane + 64 * j] = 0.f;
        if (lane == 0) a.ddepth[(size_t)b * N + n] = 0.f;
      }
      continue;
    }
    const float xf = floorf(px), yf = floorf(py);
    const int x0 = (int)xf, y0 = (int)yf;
    const float ax = px - xf, ay = py - yf;
    // ---- the 4x4 neighbourhood (minus corners) of the target map, clamped to the image: 12 coalesced row loads per
    // channel chunk; [f|gx|gy] at the 4 bilinear taps from it (grad_fixed on the fly, REFLECT rim -> 0, outside -> 0)
    float tex[CJ][4][4];       // SP: tex[j][t][e] = channel e (f, gx, gy) of the 3C map at bilinear tap t, taps clamped one by one
    if constexpr (SP) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int yy = min(max(y0 + (t >> 1), 0), H - 1), xx = min(max(x0 + (t & 1), 0), W - 1);
        const float* __restrict__ row = tgt_b + (size_t)(yy * W + xx) * 3 * C;
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
          for (int j = 0; j < CJ; ++j) tex[j][t][e] = row[e * C + (cok[j] ? lane + 64 * j : 0)];
      }
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if ((r == 0 || r == 3) && (cc == 0 || cc == 3)) continue;
        const int yy = min(max(y0 - 1 + r, 0), H - 1), xx = min(max(x0 - 1 + cc, 0), W - 1);
        const float* __restrict__ row = tgt_b + (size_t)(yy * W + xx) * C;
#pragma unroll
        for (int j = 0; j < CJ; ++j) tex[j][r][cc] = row[cok[j] ? lane + 64 * j : 0];
      }
    }
    float Sf[CJ], Sgx[CJ], Sgy[CJ], Ax[CJ][3], Ay[CJ][3], dif[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      Sf[j] = Sgx[j] = Sgy[j] = 0.f;
#pragma unroll
      for (int e = 0; e < 3; ++e) Ax[j][e] = Ay[j][e] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ix = t & 1, iy = t >> 1;
      const int tx = x0 + ix, ty = y0 + iy;
      const bool in = tx <= W - 1 && ty <= H - 1;
      const float hx = (in && tx > 0 && tx < W - 1) ? 0.5f : 0.f, hy = (in && ty > 0 && ty < H - 1) ? 0.5f : 0.f;
      const float fin = in ? 1.f : 0.f;
      const float wx = ix ? ax : 1.f - ax, wy = iy ? ay : 1.f - ay;
      const float wt = wx * wy, sx = (ix ? 1.f : -1.f) * wy, sy = (iy ? 1.f : -1.f) * wx;
#pragma unroll
      for (int j = 0; j < CJ; ++j) {
        const float F = SP ? tex[j][t][0] : fin * tex[j][1 + iy][1 + ix];
        const float GX = SP ? tex[j][t][1] : hx * (tex[j][1 + iy][2 + ix] - tex[j][1 + iy][ix]);
        const float GY = SP ? tex[j][t][2] : hy * (tex[j][2 + iy][1 + ix] - tex[j][iy][1 + ix]);
        Sf[j] = fmaf(wt, F, Sf[j]);
        Sgx[j] = fmaf(wt, GX, Sgx[j]);
        Sgy[j] = fmaf(wt, GY, Sgy[j]);
        Ax[j][0] = fmaf(sx, F, Ax[j][0]);
        Ax[j][1] = fmaf(sx, GX, Ax[j][1]);
        Ax[j][2] = fmaf(sx, GY, Ax[j][2]);
        Ay[j][0] = fmaf(sy, F, Ay[j][0]);
        Ay[j][1] = fmaf(sy, GX, Ay[j][1]);
        Ay[j][2] = fmaf(sy, GY, Ay[j][2]);
      }
    }
    float M11 = 0.f, M12 = 0.f, M22 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      if (!cok[j]) Sf[j] = Sgx[j] = Sgy[j] = f1v[j] = 0.f;
      dif[j] = f1v[j] - Sf[j];       // bundlenet.py:234
      M11 = fmaf(Sgx[j], Sgx[j], M11);
      M12 = fmaf(Sgx[j], Sgy[j], M12);
      M22 = fmaf(Sgy[j], Sgy[j], M22);
      g1 = fmaf(Sgx[j], dif[j], g1);
      g2 = fmaf(Sgy[j], dif[j], g2);
    }
    M11 = wsum(M11);
    M12 = wsum(M12);
    M22 = wsum(M22);
    g1 = wsum(g1);
    g2 = wsum(g2);
    // ---- per-pixel algebra (bundlenet.py:49-74 Jacobians with the bundle sign, J = [-Jc | jd b])
    const float iz = 1.f / Z;
    float J0[6], J1[6];
    J0[0] = fx * (-(x * y));
    J0[1] = fx * (1.f + x * x);
    J0[2] = fx * (-y);
    J0[3] = fx * iz;
    J0[4] = 0.f;
    J0[5] = fx * (-(x * iz));
    J1[0] = fy * (-1.f - y * y);
    J1[1] = fy * (x * y);
    J1[2] = fy * x;
    J1[3] = 0.f;
    J1[4] = fy * iz;
    J1[5] = fy * (-(y * iz));
    const float jd0 = fx * ((rx - rz * x) * iz), jd1 = fy * ((ry - rz * y) * iz);
    float JS0[6], JS1[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float s0 = jd0 * q[j], s1 = jd1 * q[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        s0 = fmaf(J0[i], Scc[i][j], s0);
        s1 = fmaf(J1[i], Scc[i][j], s1);
      }
      JS0[j] = s0;
      JS1[j] = s1;
    }
    float t0 = jd0 * zeta, t1 = jd1 * zeta, dM11 = 0.f, dM12 = 0.f, dM22 = 0.f, dg1 = jd0 * ee, dg2 = jd1 * ee;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      t0 = fmaf(J0[i], q[i], t0);
      t1 = fmaf(J1[i], q[i], t1);
      dM11 = fmaf(JS0[i], J0[i], dM11);
      dM12 = fmaf(JS0[i], J1[i], dM12);
      dM22 = fmaf(JS1[i], J1[i], dM22);
      dg1 = fmaf(J0[i], gbc[i], dg1);
      dg2 = fmaf(J1[i], gbc[i], dg2);
    }
    dM11 = fmaf(t0, jd0, dM11);
    dM12 = fmaf(t0, jd1, dM12);
    dM22 = fmaf(t1, jd1, dM22);
    float dJ0[6], dJ1[6], u[6];
    const float Mjd0 = M11 * jd0 + M12 * jd1, Mjd1 = M12 * jd0 + M22 * jd1;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dJ0[i] = 2.f * (M11 * JS0[i] + M12 * JS1[i]) + g1 * gbc[i];
      dJ1[i] = 2.f * (M12 * JS0[i] + M22 * JS1[i]) + g2 * gbc[i];
      u[i] = J0[i] * Mjd0 + J1[i] * Mjd1;
    }
    const float djd0 = 2.f * (M11 * t0 + M12 * t1) + g1 * ee, djd1 = 2.f * (M12 * t0 + M22 * t1) + g2 * ee;
    const float s_n = jd0 * Mjd0 + jd1 * Mjd1, r_n = jd0 * g1 + jd1 * g2;
    // ---- channel adjoints: dsrc, the 3C adjoint row, d(px, py)
    float dpx = 0.f, dpy = 0.f;
    float* __restrict__ dsrc_n = a.dsrc + ((size_t)b * N + n) * C;
    float* __restrict__ arow_n = a.arow + ((size_t)b * N + n) * 3 * C;
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      const int c = lane + 64 * j;
      if (cok[j]) {
        const float d = dif[j], gx = Sgx[j], gy = Sgy[j];
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float dd = dg1 * gx + dg2 * gy + sgn * ga[j];
        const float dgx = 2.f * (dM11 * gx + dM12 * gy) + dg1 * d;
        const float dgy = 2.f * (dM12 * gx + dM22 * gy) + dg2 * d;
        dsrc_n[c] = a.overwrite ? dd : dsrc_n[c] + dd;
        if (a.arow) {
          arow_n[c] = -dd;
          arow_n[C + c] = dgx;
          arow_n[2 * C + c] = dgy;
        }
        dpx += -dd * Ax[j][0] + dgx * Ax[j][1] + dgy * Ax[j][2];
        dpy += -dd * Ay[j][0] + dgx * Ay[j][1] + dgy * Ay[j][2];
      }
    }
    dpx = wsum(dpx);
    dpy = wsum(dpy);
    // ---- geometry adjoint
    float dx_ = fx * dpx + fx * (-y * dJ0[0] + 2.f * x * dJ0[1] - dJ0[5] * iz) + fy * (y * dJ1[1] + dJ1[2]);
    float dy_ = fy * dpy + fx * (-x * dJ0[0] - dJ0[2]) + fy * (-2.f * y * dJ1[0] + x * dJ1[1] - dJ1[5] * iz);
    float dZ_ = (fx * (-dJ0[3] + x * dJ0[5]) + fy * (-dJ1[4] + y * dJ1[5])) * iz * iz;
    float drx = fx * djd0 * iz, dry = fy * djd1 * iz, drz = -(fx * x * djd0 + fy * y * djd1) * iz;
    dx_ -= fx * rz * djd0 * iz;
    dy_ -= fy * rz * djd1 * iz;
    dZ_ -= (jd0 * djd0 + jd1 * djd1) * iz;
    const float dX = dx_ * iz, dY = dy_ * iz, dZt = dZ_ - (x * dx_ + y * dy_) * iz;
    drx = fmaf(dX, D, drx);
    dry = fmaf(dY, D, dry);
    drz = fmaf(dZt, D, drz);
    const float dD = dX * rx + dY * ry + dZt * rz;
    accT[0] += dX;
    accT[1] += dY;
    accT[2] += dZt;
    accR[0] = fmaf(drx, p0, accR[0]);
    accR[1] = fmaf(drx, p1, accR[1]);
    accR[2] = fmaf(drx, p2, accR[2]);
    accR[3] = fmaf(dry, p0, accR[3]);
    accR[4] = fmaf(dry, p1, accR[4]);
    accR[5] = fmaf(dry, p2, accR[5]);
    accR[6] = fmaf(drz, p0, accR[6]);
    accR[7] = fmaf(drz, p1, accR[7]);
    accR[8] = fmaf(drz, p2, accR[8]);
    // ---- depth, basis, coefficients
    float* __restrict__ dbas_n = a.dbasis + ((size_t)b * N + n) * K;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int k = lane + 64 * j;
      if (kok[j]) {
        float v = s_n * z2v[j] + r_n * gbd[j] + dD * wc[j];
        float su = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) su = fmaf(u[i], scd[j][i], su);
        v = fmaf(2.f, su, v);
        dbas_n[k] = a.overwrite ? v : dbas_n[k] + v;
        dwc[j] = fmaf(dD, bv[j], dwc[j]);
      }
    }
    if (lane == 0) {
      a.ddepth[(size_t)b * N + n] = a.overwrite ? dD : a.ddepth[(size_t)b * N + n] + dD;
      const int key = y0 * W + x0;
      fr[0] = __int_as_float(key);
      fr[1] = ax;
      fr[2] = ay;
      fr[3] = dg1;
      fr[4] = dg2;
      fr[5] = dM11;
      fr[6] = dM12;
      fr[7] = dM22;
      atomicAdd(&a.cnt[(size_t)b * H * W + key], 1);
    }
  }
  float* __restrict__ prow = a.part + ((size_t)b * a.G * kNumWaves + g * kNumWaves + w) * (kAdjHdr + K);
#pragma unroll
  for (int i = 0; i < 9; ++i)
    if (lane == i) prow[i] = accR[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (lane == 9 + i) prow[9 + i] = accT[i];
#pragma unroll
  for (int j = 0; j < KJ; ++j) {
    const int k = lane + 64 * j;
    if (k < K) prow[kAdjHdr + k] = dwc[j];
  }
}

// ---- per-pixel adjoint, two pixels per wave -------------------------------------------------------------------------------
// Same arithmetic as adj_pixel_kernel with a HALF wave per pixel (lanes 0-31: pixel n, lanes 32-63: pixel n + 1) and 4
// channels / coefficients per lane: every row access is one 16-byte load or store per lane, the per-pixel algebra is done once
// per pair of pixels instead of once per pixel, and the eight reductions per pixel run over 32 lanes (DPP row sum + one
// permlane swap).  C % 4 == 0, C <= 128 CJ4, K % 4 == 0, K <= 128.  Pixels outside the image take safe coordinates and skip
// their stores (the two halves of a wave diverge, so there is no early exit).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float hsum(float v) {   // sum over the 32 lanes of each half wave, every lane gets its half's total
  v = row16_sum(v);
  return bfly_merge(v, v, 16);
}

template <int CJ4, bool OW>     // OW: dsrc / ddepth / dbasis are written (BANET_ADJOINT_OVERWRITE), else read (early, with the texels) and added to
__global__ __launch_bounds__(kBlock, 2) void adj_pixel2_kernel(const AdjArgs a) {
  const banet_level_t& lv = a.lv;
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, w = wave_id();
  const int N = lv.N, C = lv.C, K = lv.K, H = lv.H, W = lv.W, P = 6 + K;
  const int hl = lane & 31, hi = lane >> 5;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ tgt_b = lv.tgt + (size_t)b * N * C;
  const float* __restrict__ bas_b = lv.basis + (size_t)b * N * K;
  // every row of the loop = a per-window base (wave-uniform: scalar registers) + an unsigned 32-bit BYTE offset (launch_dense_adjoint
  // takes this kernel only while a window's largest array, N x 3C floats, stays below 4 GB): no 64-bit address arithmetic per access
  const char* const src_c = reinterpret_cast<const char*>(src_b);
  const char* const tgt_c = reinterpret_cast<const char*>(tgt_b);
  const char* const bas_c = reinterpret_cast<const char*>(bas_b);
  const char* const arec_c = reinterpret_cast<const char*>(a.arec + (size_t)b * N * 8);
  const char* const dep_c = reinterpret_cast<const char*>(lv.depth + (size_t)b * N);
  const char* const z2_c = reinterpret_cast<const char*>(a.z2 + (size_t)b * N * K);
  char* const dsrc_c = reinterpret_cast<char*>(a.dsrc + (size_t)b * N * C);
  char* const dbas_c = reinterpret_cast<char*>(a.dbasis + (size_t)b * N * K);
  char* const ddep_c = reinterpret_cast<char*>(a.ddepth + (size_t)b * N);
  char* const frac_c = reinterpret_cast<char*>(a.frac + (size_t)b * N * kFrac);
  char* const cnt_c = reinterpret_cast<char*>(a.cnt + (size_t)b * H * W);
  char* const arow_c = a.arow ? reinterpret_cast<char*>(a.arow + (size_t)b * N * 3 * C) : nullptr;
  const unsigned rowC = (unsigned)C * 4u, rowK = (unsigned)K * 4u;
  // (pixel index x row bytes with the full-rate 24-bit multiply: N < 2^24 and the product < 2^32 are launch conditions; the 32-bit
  // v_mul_lo / v_mad_u64 the compiler takes otherwise run at a quarter of the rate, 26 of them per pixel pair)
  auto offC = [&](unsigned pix, unsigned add) { return __umul24(pix, rowC) + add; };
  auto offK = [&](unsigned pix, unsigned add) { return __umul24(pix, rowK) + add; };
  const float* __restrict__ S = a.S + (size_t)b * P * P;
  const float* __restrict__ gb = a.gb + (size_t)b * P;
  float Scc[6][6], gbc[6], Rm[9], Tv[3];      // S is symmetric (adj_sym_kernel): 21 distinct uniform values, not 36 -- the scalar registers are short
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    gbc[i] = gb[i];
#pragma unroll
    for (int j = i; j < 6; ++j) Scc[i][j] = Scc[j][i] = S[i * P + j];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) Rm[i] = a.R[b * 9 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) Tv[i] = a.T[b * 3 + i];
  const float fx0 = lv.intr[b * 4 + 0], fy0 = lv.intr[b * 4 + 1], ox0 = lv.intr[b * 4 + 2], oy0 = lv.intr[b * 4 + 3];
  const float fx = fx0 / lv.scale, fy = fy0 / lv.scale, ox = ox0 / lv.scale, oy = oy0 / lv.scale;
  // this lane's 4 depth coefficients (k = 4 hl .. 4 hl + 3) and channel quads (c = 4 hl + 128 j)
  const int k0 = 4 * hl;
  const bool kok = k0 < K;
  f32x4 wc = {0.f, 0.f, 0.f, 0.f}, gbd = {0.f, 0.f, 0.f, 0.f}, dwc = {0.f, 0.f, 0.f, 0.f};
  f32x4 scd[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) scd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (kok) {
    wc = *reinterpret_cast<const f32x4*>(a.Wc + (size_t)b * K + k0);
#pragma unroll
    for (int e = 0; e < 4; ++e) gbd[e] = gb[6 + k0 + e];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) scd[i][e] = S[i * P + 6 + k0 + e];
  }
  bool cok[CJ4];
  unsigned chb[CJ4];      // this lane's channel quad of chunk j as a byte offset into a row (clamped: a lane without channels reads quad 0)
  f32x4 ga[CJ4];
#pragma unroll
  for (int j = 0; j < CJ4; ++j) {
    const int c = 4 * hl + 128 * j;
    cok[j] = c < C;
    chb[j] = cok[j] ? (unsigned)c * 4u : 0u;
    ga[j] = cok[j] ? *reinterpret_cast<const f32x4*>(a.gabs + (size_t)b * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float accR[9], accT[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) accR[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) accT[i] = 0.f;

  const int nw = a.G * kNumWaves, chunk = ((N + nw - 1) / nw + 1) & ~1;   // even: the pairs of a wave never straddle two waves
  const int n_lo = (g * kNumWaves + w) * chunk, n_hi = min(N, n_lo + chunk);
  // Software pipeline (round 6; by ablation the kernel was bound by two exposed memory round trips per pixel pair, not by its
  // bytes): the rows a pair's geometry needs -- basis, depth, the (q, zeta, e) record -- are requested one pair ahead; the 12 target
  // texel loads go out as soon as the projection is known and the part of the per-pixel algebra that does not depend on the texels
  // (Jacobians, J S_cc + jd q^T, t, dM, dg) runs while they travel.
  auto pix_of = [&](int n2) { const int n = n2 + hi; return n < n_hi ? n : n_hi - 1; };     // a dead upper half recomputes the lower pixel
  const int kc = kok ? k0 : 0;      // loads are unconditional at clamped offsets (a conditional load is merged with its default by
                                    // copies that wait for it on the spot); what a lane without coefficients / channels reads is masked where it is used
  f32x4 bv_n, ar0_n, ar1_n;
  float dep_n;
  const unsigned kcb = (unsigned)kc * 4u;
  {
    const unsigned q0 = (unsigned)pix_of(n_lo < n_hi ? n_lo : 0);
    bv_n = *reinterpret_cast<const f32x4*>(bas_c + offK(q0, kcb));
    ar0_n = *reinterpret_cast<const f32x4*>(arec_c + q0 * 32u);
    ar1_n = *reinterpret_cast<const f32x4*>(arec_c + (q0 * 32u + 16u));
    dep_n = *reinterpret_cast<const float*>(dep_c + q0 * 4u);
  }
  int by = n_lo / W, bx = n_lo - by * W;               // (column, row) of the pair's first pixel, stepped along with n2 (uniform)
  for (int n2 = n_lo; n2 < n_hi; n2 += 2) {
    const int n = n2 + hi;
    const bool live = n < n_hi;
    const int nn = live ? n : n_hi - 1;                 // a dead upper half recomputes the lower pixel (n_hi - 1 = n2: chunks are even) and stores nothing
    int qx = bx + (live ? hi : 0), qy = by;
    if (qx >= W) {
      qx = 0;
      ++qy;
    }
    float p0 = ((float)qx * lv.scale - ox0) / fx0, p1 = ((float)qy * lv.scale - oy0) / fy0, p2 = 1.f;
    if (lv.normalize_rays) {
      const float inv = 1.f / sqrtf(fmaxf(p0 * p0 + p1 * p1 + p2 * p2, 1e-12f));
      p0 *= inv;
      p1 *= inv;
      p2 *= inv;
    }
    const unsigned qn0 = (unsigned)nn;
    const f32x4 bv = bv_n, ar0 = ar0_n, ar1 = ar1_n;
    const float qv[6] = {ar0[0], ar0[1], ar0[2], ar0[3], ar1[0], ar1[1]};
    const float zeta = ar1[2], ee = ar1[3];
    const float D = dep_n + hsum(bv[0] * wc[0] + bv[1] * wc[1] + bv[2] * wc[2] + bv[3] * wc[3]);
    const float rx = Rm[0] * p0 + Rm[1] * p1 + Rm[2] * p2;
    const float ry = Rm[3] * p0 + Rm[4] * p1 + Rm[5] * p2;
    const float rz = Rm[6] * p0 + Rm[7] * p1 + Rm[8] * p2;
    const float X = rx * D + Tv[0], Y = ry * D + Tv[1], Z0 = rz * D + Tv[2];
    float x = X / Z0, y = Y / Z0, Z = Z0;
    float px = fx * x + ox, py = fy * y + oy;
    const bool m = live && (px >= 0.f) && (px <= (float)(W - 1)) && (py >= 0.f) && (py <= (float)(H - 1));
    if (!m) {            // safe stand-ins: everything below stays finite, nothing of it is stored
      x = 0.f;
      y = 0.f;
      Z = 1.f;
      px = 1.5f;
      py = 1.5f;
    }
    const float xf = floorf(px), yf = floorf(py);
    const int x0 = (int)xf, y0 = (int)yf;
    const float ax = px - xf, ay = py - yf;
    // ---- the 4x4 neighbourhood (minus corners), clamped: 12 row loads of 16 bytes per lane and channel chunk
    f32x4 tex[CJ4][4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if ((r == 0 || r == 3) && (cc == 0 || cc == 3)) continue;
        const int yy = min(max(y0 - 1 + r, 0), H - 1), xx = min(max(x0 - 1 + cc, 0), W - 1);
        const unsigned row = offC((unsigned)(__mul24(yy, W) + xx), 0u);
#pragma unroll
        for (int j = 0; j < CJ4; ++j) tex[j][r][cc] = *reinterpret_cast<const f32x4*>(tgt_c + (row + chb[j]));
      }
    // this pair's source row and z2 row (needed after the taps / at the very end), the next pair's geometry rows
    const f32x4 z2v = *reinterpret_cast<const f32x4*>(z2_c + offK(qn0, kcb));
    f32x4 f1v[CJ4];
#pragma unroll
    for (int j = 0; j < CJ4; ++j) f1v[j] = *reinterpret_cast<const f32x4*>(src_c + offC(qn0, chb[j]));
    f32x4 ds_old[CJ4], db_old;      // accumulate mode: the old contents travel with the texels
    float dd_old = 0.f;
    if constexpr (!OW) {
#pragma unroll
      for (int j = 0; j < CJ4; ++j) ds_old[j] = *reinterpret_cast<const f32x4*>(dsrc_c + offC(qn0, chb[j]));
      db_old = *reinterpret_cast<const f32x4*>(dbas_c + offK(qn0, kcb));
      dd_old = *reinterpret_cast<const float*>(ddep_c + qn0 * 4u);
    }
    {
      const unsigned nx = (unsigned)pix_of(n2 + 2 < n_hi ? n2 + 2 : n2);      // (the last pair requests itself again: no branch around the loads)
      bv_n = *reinterpret_cast<const f32x4*>(bas_c + offK(nx, kcb));
      ar0_n = *reinterpret_cast<const f32x4*>(arec_c + nx * 32u);
      ar1_n = *reinterpret_cast<const f32x4*>(arec_c + (nx * 32u + 16u));
      dep_n = *reinterpret_cast<const float*>(dep_c + nx * 4u);
    }
    // ---- per-pixel algebra, the part without M and g (as adj_pixel_kernel), under the texel loads
    // (written on PAIRS (row 0, row 1) of the 2 x 6 Jacobian: one packed operation per pair, element by element the same chain of
    // fused multiply-adds as adj_pixel_kernel -- left to itself the compiler packs across the column index instead and spends
    // ~130 register copies per pixel pair on building its operands)
    const float iz = 1.f / Z;
    const f32x2 fxy = {fx, fy};
    f32x2 Jp[6];
    Jp[0] = fxy * f32x2{-(x * y), -1.f - y * y};
    Jp[1] = fxy * f32x2{1.f + x * x, x * y};
    Jp[2] = fxy * f32x2{-y, x};
    Jp[3] = f32x2{fx * iz, 0.f};
    Jp[4] = f32x2{0.f, fy * iz};
    Jp[5] = fxy * f32x2{-(x * iz), -(y * iz)};
    const f32x2 jdp = fxy * f32x2{(rx - rz * x) * iz, (ry - rz * y) * iz};
    const float jd0 = jdp[0], jd1 = jdp[1];
    f32x2 JSp[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x2 sp = jdp * qv[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) sp = fma2(Jp[i], splat2(Scc[i][j]), sp);
      JSp[j] = sp;
    }
    f32x2 tp = jdp * zeta, dgp = jdp * ee, dMd = {0.f, 0.f};
    float dM12 = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      tp = fma2(Jp[i], splat2(qv[i]), tp);
      dMd = fma2(JSp[i], Jp[i], dMd);
      dM12 = fmaf(JSp[i][0], Jp[i][1], dM12);
      dgp = fma2(Jp[i], splat2(gbc[i]), dgp);
    }
    dMd = fma2(tp, jdp, dMd);
    dM12 = fmaf(tp[0], jd1, dM12);
    const float t0 = tp[0], t1 = tp[1], dg1 = dgp[0], dg2 = dgp[1], dM11 = dMd[0], dM22 = dMd[1];
    asm volatile("" ::: "memory");      // (keeps the block above ahead of the first use of a texel)
    f32x4 Sf[CJ4], Sgx[CJ4], Sgy[CJ4], Ax[CJ4][3], Ay[CJ4][3];
#pragma unroll
    for (int j = 0; j < CJ4; ++j) {
      Sf[j] = Sgx[j] = Sgy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 3; ++e) Ax[j][e] = Ay[j][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the four taps.  `inner` (every footprint of the wave off the image rim: nearly always) knows fin = 1, hx = hy = 0.5 at compile
    // time -- ~75 instructions less per pair; same values, and explicit fused multiply-adds so that the two paths (and the two
    // instantiations of this kernel) round alike
    auto taps = [&](auto innerc) __attribute__((always_inline)) {
      constexpr bool IN = decltype(innerc)::value;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ix = t & 1, iy = t >> 1;
        const int tx = x0 + ix, ty = y0 + iy;
        const bool in = IN || (tx <= W - 1 && ty <= H - 1);
        const float hx = (IN || (in && tx > 0 && tx < W - 1)) ? 0.5f : 0.f, hy = (IN || (in && ty > 0 && ty < H - 1)) ? 0.5f : 0.f;
        const float wx = ix ? ax : 1.f - ax, wy = iy ? ay : 1.f - ay;
        const float wt = wx * wy, sx = (ix ? 1.f : -1.f) * wy, sy = (iy ? 1.f : -1.f) * wx;
        const f32x4 wt4 = {wt, wt, wt, wt}, sx4 = {sx, sx, sx, sx}, sy4 = {sy, sy, sy, sy};
#pragma unroll
        for (int j = 0; j < CJ4; ++j) {
          f32x4 F = tex[j][1 + iy][1 + ix];
          if (!IN && !in) F = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 GX = hx * (tex[j][1 + iy][2 + ix] - tex[j][1 + iy][ix]);
          const f32x4 GY = hy * (tex[j][2 + iy][1 + ix] - tex[j][iy][1 + ix]);
          Sf[j] = fma4(wt4, F, Sf[j]);
          Sgx[j] = fma4(wt4, GX, Sgx[j]);
          Sgy[j] = fma4(wt4, GY, Sgy[j]);
          Ax[j][0] = fma4(sx4, F, Ax[j][0]);
          Ax[j][1] = fma4(sx4, GX, Ax[j][1]);
          Ax[j][2] = fma4(sx4, GY, Ax[j][2]);
          Ay[j][0] = fma4(sy4, F, Ay[j][0]);
          Ay[j][1] = fma4(sy4, GX, Ay[j][1]);
          Ay[j][2] = fma4(sy4, GY, Ay[j][2]);
        }
      }
    };
    if (__builtin_expect(__all(x0 >= 1 && y0 >= 1 && x0 + 2 <= W - 1 && y0 + 2 <= H - 1) != 0, 1)) {
      taps(std::true_type{});
    } else {
      asm volatile("" ::: "memory");
      taps(std::false_type{});
      asm volatile("" ::: "memory");
    }
    f32x4 dif[CJ4];
    float M11 = 0.f, M12 = 0.f, M22 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int j = 0; j < CJ4; ++j) {
      if (!cok[j]) Sf[j] = Sgx[j] = Sgy[j] = f1v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      dif[j] = f1v[j] - Sf[j];       // bundlenet.py:234
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        M11 = fmaf(Sgx[j][e], Sgx[j][e], M11);
        M12 = fmaf(Sgx[j][e], Sgy[j][e], M12);
        M22 = fmaf(Sgy[j][e], Sgy[j][e], M22);
        g1 = fmaf(Sgx[j][e], dif[j][e], g1);
        g2 = fmaf(Sgy[j][e], dif[j][e], g2);
      }
    }
    M11 = hsum(M11);
    M12 = hsum(M12);
    M22 = hsum(M22);
    g1 = hsum(g1);
    g2 = hsum(g2);
    // ---- per-pixel algebra, the part with M and g
    f32x2 dJp[6];
    float u[6];
    const f32x2 Mr0 = {M11, M12}, Mr1 = {M12, M22}, gp = {g1, g2};
    const float Mjd0 = fmaf(M11, jd0, M12 * jd1), Mjd1 = fmaf(M12, jd0, M22 * jd1);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dJp[i] = fma2(splat2(2.f), fma2(Mr0, splat2(JSp[i][0]), Mr1 * JSp[i][1]), gp * gbc[i]);
      u[i] = fmaf(Jp[i][0], Mjd0, Jp[i][1] * Mjd1);
    }
    const f32x2 djdp = fma2(splat2(2.f), fma2(Mr0, splat2(t0), Mr1 * t1), gp * ee);
    const float djd0 = djdp[0], djd1 = djdp[1];
    const float s_n = fmaf(jd0, Mjd0, jd1 * Mjd1), r_n = fmaf(jd0, g1, jd1 * g2);
    float dJ0[6], dJ1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dJ0[i] = dJp[i][0];
      dJ1[i] = dJp[i][1];
    }
    // ---- channel adjoints
    float dpx = 0.f, dpy = 0.f;
    const unsigned dsrc_o = offC(qn0, (unsigned)hl * 16u), arow_o = 3u * offC(qn0, 0u) + (unsigned)hl * 16u;     // (base + 32-bit offset at every store)
#pragma unroll
    for (int j = 0; j < CJ4; ++j) {
      if (cok[j]) {
        f32x4 dd, dgx, dgy;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = dif[j][e], gx = Sgx[j][e], gy = Sgy[j][e];
          const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
          // (explicit fused operations: the two instantiations of this kernel -- write / accumulate -- must round alike, and the
          // compiler's own contraction of a * b + c * d + e * f is free to differ between them)
          dd[e] = fmaf(sgn, ga[j][e], fmaf(dg2, gy, dg1 * gx));
          dgx[e] = fmaf(dg1, d, 2.f * fmaf(dM12, gy, dM11 * gx));
          dgy[e] = fmaf(dg2, d, 2.f * fmaf(dM22, gy, dM12 * gx));
          dpx += -dd[e] * Ax[j][0][e] + dgx[e] * Ax[j][1][e] + dgy[e] * Ax[j][2][e];
          dpy += -dd[e] * Ay[j][0][e] + dgx[e] * Ay[j][1][e] + dgy[e] * Ay[j][2][e];
        }
        if (OW && live && !m) *reinterpret_cast<f32x4*>(dsrc_c + (dsrc_o + 512u * j)) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m) {
          f32x4* ds = reinterpret_cast<f32x4*>(dsrc_c + (dsrc_o + 512u * j));
          if constexpr (OW)
            *ds = dd;
          else
            *ds = ds_old[j] + dd;
          if (arow_c) {
            *reinterpret_cast<f32x4*>(arow_c + (arow_o + 512u * j)) = -dd;
            *reinterpret_cast<f32x4*>(arow_c + (arow_o + rowC + 512u * j)) = dgx;
            *reinterpret_cast<f32x4*>(arow_c + (arow_o + 2u * rowC + 512u * j)) = dgy;
          }
        }
      }
    }
    dpx = hsum(dpx);
    dpy = hsum(dpy);
    // ---- geometry adjoint
    float dx_ = fx * dpx + fx * (-y * dJ0[0] + 2.f * x * dJ0[1] - dJ0[5] * iz) + fy * (y * dJ1[1] + dJ1[2]);
    float dy_ = fy * dpy + fx * (-x * dJ0[0] - dJ0[2]) + fy * (-2.f * y * dJ1[0] + x * dJ1[1] - dJ1[5] * iz);
    float dZ_ = (fx * (-dJ0[3] + x * dJ0[5]) + fy * (-dJ1[4] + y * dJ1[5])) * iz * iz;
    float drx = fx * djd0 * iz, dry = fy * djd1 * iz, drz = -(fx * x * djd0 + fy * y * djd1) * iz;
    dx_ -= fx * rz * djd0 * iz;
    dy_ -= fy * rz * djd1 * iz;
    dZ_ -= (jd0 * djd0 + jd1 * djd1) * iz;
    const float ms = m ? 1.f : 0.f;
    const float dX = ms * dx_ * iz, dY = ms * dy_ * iz, dZt = ms * (dZ_ - (x * dx_ + y * dy_) * iz);
    drx = fmaf(dX, D, ms * drx);
    dry = fmaf(dY, D, ms * dry);
    drz = fmaf(dZt, D, ms * drz);
    const float dD = dX * rx + dY * ry + dZt * rz;
    accT[0] += dX;
    accT[1] += dY;
    accT[2] += dZt;
    accR[0] = fmaf(drx, p0, accR[0]);
    accR[1] = fmaf(drx, p1, accR[1]);
    accR[2] = fmaf(drx, p2, accR[2]);
    accR[3] = fmaf(dry, p0, accR[3]);
    accR[4] = fmaf(dry, p1, accR[4]);
    accR[5] = fmaf(dry, p2, accR[5]);
    accR[6] = fmaf(drz, p0, accR[6]);
    accR[7] = fmaf(drz, p1, accR[7]);
    accR[8] = fmaf(drz, p2, accR[8]);
    // ---- depth, basis, coefficients
    if (m && kok) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float su = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) su = fmaf(u[i], scd[i][e], su);
        v[e] = fmaf(2.f, su, fmaf(dD, wc[e], fmaf(r_n, gbd[e], s_n * z2v[e])));
        dwc[e] = fmaf(dD, bv[e], dwc[e]);
      }
      f32x4* db = reinterpret_cast<f32x4*>(dbas_c + offK(qn0, kcb));
      if constexpr (OW)
        *db = v;
      else
        *db = db_old + v;
    } else if (OW && live && kok) {
      *reinterpret_cast<f32x4*>(dbas_c + offK(qn0, kcb)) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (hl == 0 && live) {
      const unsigned fro = qn0 * (unsigned)(kFrac * 4), ddo = qn0 * 4u;
      if (OW && !m) *reinterpret_cast<float*>(ddep_c + ddo) = 0.f;
      if (m) {
        *reinterpret_cast<float*>(ddep_c + ddo) = OW ? dD : dd_old + dD;
        const int key = y0 * W + x0;
        *reinterpret_cast<f32x4*>(frac_c + fro) = f32x4{__int_as_float(key), ax, ay, dg1};
        *reinterpret_cast<f32x4*>(frac_c + (fro + 16u)) = f32x4{dg2, dM11, dM12, dM22};
        atomicAdd(reinterpret_cast<int*>(cnt_c + (unsigned)key * 4u), 1);
      } else {
        *reinterpret_cast<float*>(frac_c + fro) = __int_as_float(-1);
      }
    }
    bx += 2;
    while (bx >= W) {
      bx -= W;
      ++by;
    }
  }
  // the wave's partial row: lower half + upper half, fixed order
  float* __restrict__ prow = a.part + ((size_t)b * a.G * kNumWaves + g * kNumWaves + w) * (kAdjHdr + K);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const float v = bfly_merge(accR[i], accR[i], 32);
    if (lane == i) prow[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float v = bfly_merge(accT[i], accT[i], 32);
    if (lane == 9 + i) prow[9 + i] = v;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = bfly_merge(dwc[e], dwc[e], 32);
    if (hi == 0 && kok) prow[kAdjHdr + k0 + e] = v;
  }
}

__global__ void adj_fold_kernel(const float* __restrict__ part, int rows, int K, float* __restrict__ dpose) {
  const int b = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 12 + K) return;
  const int off = e < 12 ? e : kAdjHdr + (e - 12);
  const float* p = part + (size_t)b * rows * (kAdjHdr + K) + off;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = 0;
  for (; i + 3 < rows; i += 4) {
    s0 += p[(size_t)(i + 0) * (kAdjHdr + K)];
    s1 += p[(size_t)(i + 1) * (kAdjHdr + K)];
    s2 += p[(size_t)(i + 2) * (kAdjHdr + K)];
    s3 += p[(size_t)(i + 3) * (kAdjHdr + K)];
  }
  for (; i < rows; ++i) s0 += p[(size_t)i * (kAdjHdr + K)];
  dpose[(size_t)b * (12 + K) + e] = (s0 + s1) + (s2 + s3);
}

// ---- target cell lists ---------------------------------------------------------------------------------------------
// Exclusive scan of the per-cell counts of every window, in three small launches (chunk sums -> scan of the chunk sums -> scan
// inside the chunks): 1024 cells per 256-thread workgroup.  (A one-workgroup-per-window scan took 0.76 ms at 640x480.)
constexpr int kScanChunk = 1024;

__global__ __launch_bounds__(256) void adj_scan_sums_kernel(const int* __restrict__ cnt, int* __restrict__ chunk_sum, int HW, int nchunks) {
  __shared__ int sRed[4];
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const int* __restrict__ c = cnt + (size_t)b * HW;
  int s = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = ch * kScanChunk + 4 * tid + e;
    s += i < HW ? c[i] : 0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((tid & 63) == 0) sRed[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) chunk_sum[(size_t)b * nchunks + ch] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
}

__global__ __launch_bounds__(kScanThreads) void adj_scan_offsets_kernel(int* __restrict__ chunk_sum, int nchunks) {
  __shared__ int sSum[kScanThreads];
  const int b = blockIdx.x, tid = threadIdx.x;
  int* __restrict__ c = chunk_sum + (size_t)b * nchunks;
  int carry = 0;
  for (int base = 0; base < nchunks; base += kScanThreads) {     // in place: chunk sums -> exclusive offsets
    const int i = base + tid;
    const int v = i < nchunks ? c[i] : 0;
    sSum[tid] = v;
    __syncthreads();
    for (int d = 1; d < kScanThreads; d <<= 1) {
      const int t = tid >= d ? sSum[tid - d] : 0;
      __syncthreads();
      sSum[tid] += t;
      __syncthreads();
    }
    if (i < nchunks) c[i] = carry + sSum[tid] - v;
    carry += sSum[kScanThreads - 1];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void adj_scan_apply_kernel(const int* __restrict__ cnt, const int* __restrict__ chunk_off,
                                                             int* __restrict__ start, int* __restrict__ cursor, int HW, int nchunks) {
  __shared__ int sSum[256];
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const int* __restrict__ c = cnt + (size_t)b * HW;
  int v[4], s = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = ch * kScanChunk + 4 * tid + e;
    v[e] = i < HW ? c[i] : 0;
    s += v[e];
  }
  sSum[tid] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int t = tid >= d ? sSum[tid - d] : 0;
    __syncthreads();
    sSum[tid] += t;
    __syncthreads();
  }
  int run = chunk_off[(size_t)b * nchunks + ch] + sSum[tid] - s;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = ch * kScanChunk + 4 * tid + e;
    if (i < HW) {
      start[2 * ((size_t)b * HW + i)] = run;        // (start, count) pairs: one 8-byte load per cell in adj_map_kernel
      start[2 * ((size_t)b * HW + i) + 1] = v[e];
      cursor[(size_t)b * HW + i] = run;
    }
    run += v[e];
  }
}

static void launch_cell_scan(const int* cnt, int* start, int* cursor, int* chunk_tmp, int B, int HW, hipStream_t s) {
  const int nchunks = (HW + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(adj_scan_sums_kernel, dim3(nchunks, B), dim3(256), 0, s, cnt, chunk_tmp, HW, nchunks);
  hipLaunchKernelGGL(adj_scan_offsets_kernel, dim3(B), dim3(kScanThreads), 0, s, chunk_tmp, nchunks);
  hipLaunchKernelGGL(adj_scan_apply_kernel, dim3(nchunks, B), dim3(256), 0, s, cnt, chunk_tmp, start, cursor, HW, nchunks);
}

__global__ void adj_fill_kernel(const float* __restrict__ frac, int* __restrict__ cursor, int* __restrict__ list, int N, int HW) {
  const int b = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int key = __float_as_int(frac[((size_t)b * N + n) * kFrac]);
  if (key < 0) return;
  const int slot = atomicAdd(&cursor[(size_t)b * HW + key], 1);   // the consumer orders each cell's entries itself
  list[(size_t)b * N + slot] = n;
}

// ---- round 6: the target map's gradient per 8x8 texel tile, without the 3C rows in memory -------------------------------
// Until round 5 every pixel wrote its 3C adjoint row of the sampled [f|gx|gy] vector (1.5 KB), adj_map2_kernel gathered four of
// them per texel into the [f|gx|gy] map adjoint dmap3 (another 1.5 KB per texel, read-modify-write over the iterations) and
// target_map_adjoint4_kernel folded grad_fixed^T once per level: 8+ KB of traffic per pixel and iteration, 30 GB of buffers at 32
// windows of 640x480.  Here ONE wave owns a tile of TW x TH target texels exclusively: it walks the cell lists of the tile and of
// the ring around it ((TW + 3) x (TH + 3) cells: every pixel whose 12-texel footprint -- four bilinear taps plus their
// central-difference neighbours -- touches the tile), recomputes the pixel's channel adjoints (-dd, dgx, dgy) from its source row,
// the 12 target texels of its cell and the eight per-pixel scalars adj_pixel*_kernel left in its record, and accumulates
// grad_fixed^T of them (bundlenet.py:92-100, REFLECT rim -> 0) straight into the tile's accumulators in LDS, in a fixed order
// (cells row-major, ascending pixel index inside a cell): no float atomics between waves, bit-reproducible, and the only thing
// written is the target map's gradient itself -- once per tile.
constexpr int kSortSerial = 32;     // cells with up to this many pixels are ordered by one thread (insertion sort)

// One thread per target cell: order the cell's pixel list (adj_fill_kernel's slots come from an atomic cursor) and write the
// records in list order, key replaced by the pixel index.  Fuller cells (a collapsed warp) go to the big-cell queue.
__global__ void adj_cellsort_kernel(const int2* __restrict__ cs, int* __restrict__ list, const float* __restrict__ frac,
                                    float* __restrict__ lrec, int2* __restrict__ lidx, int* __restrict__ bigq, int N, int HW, int W,
                                    int bigq_cap) {
  const int b = blockIdx.y, cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= HW) return;
  const int2 v = cs[(size_t)b * HW + cell];
  const int s0 = v.x, L = v.y;
  if (L == 0) return;
  int* __restrict__ li = list + (size_t)b * N + s0;
  if (L > kSortSerial) {
    const int slot = atomicAdd(&bigq[0], 1);
    if (slot < bigq_cap) bigq[1 + slot] = b * HW + cell;     // (cap = every possible big cell: never exceeded)
    return;
  }
  for (int i = 1; i < L; ++i) {
    const int key = li[i];
    int j = i - 1;
    while (j >= 0 && li[j] > key) {
      li[j + 1] = li[j];
      --j;
    }
    li[j + 1] = key;
  }
  const float* __restrict__ fr = frac + (size_t)b * N * kFrac;
  float* __restrict__ lr = lrec + ((size_t)b * N + s0) * kFrac;
  int2* __restrict__ lx = lidx + (size_t)b * N + s0;
  const int cy = cell / W, cxy = (cell - cy * W) | (cy << 16);      // (W, H < 65536)
  for (int i = 0; i < L; ++i) {
    const int n = li[i];
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(fr + (size_t)n * kFrac), r1 = *reinterpret_cast<const f32x4*>(fr + (size_t)n * kFrac + 4);
    *reinterpret_cast<f32x4*>(lr + (size_t)i * kFrac) = f32x4{__int_as_float(n), r0[1], r0[2], r0[3]};
    *reinterpret_cast<f32x4*>(lr + (size_t)i * kFrac + 4) = r1;
    lx[i] = make_int2(n, cxy);
  }
}

// One workgroup per queued cell: rank sort through list2 (pixel indices are distinct), then the records.  O(L^2 / 256) -- only a
// degenerate warp (hundreds of pixels in one texel cell) gets here; adj_map_kernel's per-texel selection had the same order.
__global__ __launch_bounds__(256) void adj_bigcell_kernel(const int2* __restrict__ cs, int* __restrict__ list, int* __restrict__ list2,
                                                          const float* __restrict__ frac, float* __restrict__ lrec,
                                                          int2* __restrict__ lidx, const int* __restrict__ bigq, int N, int HW, int W,
                                                          int bigq_cap) {
  const int nbig = min(bigq[0], bigq_cap);
  for (int qi = blockIdx.x; qi < nbig; qi += gridDim.x) {
    const int gc = bigq[1 + qi], b = gc / HW;
    const int2 v = cs[gc];
    const int s0 = v.x, L = v.y;
    int* __restrict__ li = list + (size_t)b * N + s0;
    int* __restrict__ lo = list2 + (size_t)b * N + s0;
    for (int e = threadIdx.x; e < L; e += 256) {
      const int me = li[e];
      int rank = 0;
      for (int j = 0; j < L; ++j) rank += li[j] < me ? 1 : 0;
      lo[rank] = me;
    }
    __syncthreads();
    const float* __restrict__ fr = frac + (size_t)b * N * kFrac;
    float* __restrict__ lr = lrec + ((size_t)b * N + s0) * kFrac;
    int2* __restrict__ lx = lidx + (size_t)b * N + s0;
    const int cell = gc - b * HW, cy = cell / W, cxy = (cell - cy * W) | (cy << 16);
    for (int e = threadIdx.x; e < L; e += 256) {
      const int n = lo[e];
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(fr + (size_t)n * kFrac), r1 = *reinterpret_cast<const f32x4*>(fr + (size_t)n * kFrac + 4);
      *reinterpret_cast<f32x4*>(lr + (size_t)e * kFrac) = f32x4{__int_as_float(n), r0[1], r0[2], r0[3]};
      *reinterpret_cast<f32x4*>(lr + (size_t)e * kFrac + 4) = r1;
      lx[e] = make_int2(n, cxy);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L; e += 256) li[e] = lo[e];     // (the list itself in order too: diagnostics, the old per-texel path)
    __syncthreads();
  }
}

template <int V> struct VecOf;
template <> struct VecOf<1> { typedef float T; };
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <int V> __device__ __forceinline__ typename VecOf<V>::T vsplat(float x);
template <> __device__ __forceinline__ float vsplat<1>(float x) { return x; }
template <> __device__ __forceinline__ VecOf<2>::T vsplat<2>(float x) { return VecOf<2>::T{x, x}; }

// CJ x V x 64 >= C: lane owns the V consecutive channels V lane + 64 V j (+ e); V = 2 needs an even C.  One wave per workgroup;
// LDS = (TW TH + 1) CJ V 64 floats, the V accumulators of (texel t, chunk j) of this lane side by side at ((t CJ + j) 64 + lane) V
// (8-byte accesses, conflict-free); slot TW TH is a dummy that takes the footprint texels outside the tile (no branch per
// destination).
// Work items are numbered so that the eight XCDs (workgroup id mod 8) each walk a contiguous range of tiles: the target halo and the
// source rows that neighbouring tiles share then meet in one L2.
// Memory pipeline: the record two pixels ahead, the source row one pixel ahead and the 12 target texels of the next non-empty cell
// are requested at the top of an iteration and a full s_waitcnt vmcnt(0) closes it (after ~300 instructions of arithmetic): the
// loop header then has nothing pending, so the compiler's conservative wait counts at the loop join cannot stall on loads that
// were only just issued (the first version of this kernel did exactly that: 3.4 us per pixel).
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), expcnt / lgkmcnt untouched

template <int CJ, int V, int TW, int TH>
__global__ __launch_bounds__(64) void adj_tile_kernel(const AdjArgs a, int tiles_x, int ntiles, int total, int chunk) {
  typedef typename VecOf<V>::T vec;
  static_assert((TW + 3) * (TH + 3) <= 128, "the tile's cell table is two entries per lane");
  constexpr int kSlotV = CJ * 64;               // vecs per texel slot: accumulator of (texel t, chunk j) of this lane at (t CJ + j) 64 + lane
  extern __shared__ __attribute__((aligned(16))) float sAcc[];
  vec* sAccV = reinterpret_cast<vec*>(sAcc);
  const int lane = threadIdx.x;
  const int kk = blockIdx.x >> 3, work = (blockIdx.x & 7) * chunk + kk;
  if (kk >= chunk || work >= total) return;
  const int b = work / ntiles, tile = work - b * ntiles;
  const int tyi = tile / tiles_x, tx0 = (tile - tyi * tiles_x) * TW, ty0 = tyi * TH;
  const int N = a.lv.N, C = a.lv.C, H = a.lv.H, W = a.lv.W, HW = H * W;
  const float* __restrict__ src_b = a.lv.src + (size_t)b * N * C;
  const float* __restrict__ tgt_b = a.lv.tgt + (size_t)b * HW * C;
  const float* __restrict__ lrec = a.lrec + (size_t)b * N * kFrac;
  const int2* __restrict__ cs = reinterpret_cast<const int2*>(a.start) + (size_t)b * HW;
  bool cok[CJ];
  int coff[CJ];
  vec ga[CJ];
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int c = V * lane + 64 * V * j;
    cok[j] = c < C;
    coff[j] = cok[j] ? c : 0;
    ga[j] = *reinterpret_cast<const vec*>(a.gabs + (size_t)b * C + coff[j]);
    if (!cok[j]) ga[j] = vsplat<V>(0.f);
  }
  for (int i = lane; i < (TW * TH + 1) * kSlotV; i += 64) sAccV[i] = vsplat<V>(0.f);

  // the cells whose pixels can touch the tile: x0 in [tx0 - 2, tx0 + TW], y0 in [ty0 - 2, ty0 + TH], clipped to the image
  const int cxa = max(tx0 - 2, 0), cxb = min(tx0 + TW, W - 1), ncx = cxb - cxa + 1;
  const int cya = max(ty0 - 2, 0), cyb = min(ty0 + TH, H - 1), ncy = cyb - cya + 1, ncell = ncx * ncy;
  int cst[2], ccn[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int id = lane + 64 * e;
    const bool ok = id < ncell;
    const int r = id / ncx, c = id - r * ncx;
    const int2 v = cs[ok ? (cya + r) * W + cxa + c : 0];
    cst[e] = v.x;
    ccn[e] = ok ? v.y : 0;
  }
  unsigned long long q0 = __ballot(ccn[0] > 0), q1 = __ballot(ccn[1] > 0);
  auto next_cell = [&]() -> int {      // pops the id of the next non-empty cell in row-major order, -1 = none
    int id = -1;
    if (q0) {
      id = __builtin_ctzll(q0);
      q0 &= q0 - 1;
    } else if (q1) {
      id = 64 + __builtin_ctzll(q1);
      q1 &= q1 - 1;
    }
    return id;
  };
  auto cell_range = [&](int id, int& s, int& L) {
    const int l = id & 63;
    s = id < 64 ? __builtin_amdgcn_readlane(cst[0], l) : __builtin_amdgcn_readlane(cst[1], l);
    L = id < 64 ? __builtin_amdgcn_readlane(ccn[0], l) : __builtin_amdgcn_readlane(ccn[1], l);
  };
  const unsigned rowB = (unsigned)W * (unsigned)C, laneoff = 0;     // floats per target row
  (void)laneoff;
  auto load_tex = [&](int id, vec (&tx)[4][4][CJ]) {
    const int r0 = id / ncx, cx = cxa + id - r0 * ncx, cy = cya + r0;
    unsigned xo[4], yo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xo[k] = (unsigned)min(max(cx - 1 + k, 0), W - 1) * (unsigned)C;
      yo[k] = (unsigned)min(max(cy - 1 + k, 0), H - 1) * rowB;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if ((r == 0 || r == 3) && (cc == 0 || cc == 3)) continue;
        const float* __restrict__ row = tgt_b + (yo[r] + xo[cc]);      // (a window's target map is < 2^32 floats: plan check)
#pragma unroll
        for (int j = 0; j < CJ; ++j) tx[r][cc][j] = *reinterpret_cast<const vec*>(row + coff[j]);
      }
  };
  auto load_rec = [&](int i, f32x4& r0, f32x4& r1) {
    r0 = *reinterpret_cast<const f32x4*>(lrec + (size_t)i * kFrac);
    r1 = *reinterpret_cast<const f32x4*>(lrec + (size_t)i * kFrac + 4);
  };
  auto load_src = [&](int n, vec (&f)[CJ]) {
#pragma unroll
    for (int j = 0; j < CJ; ++j) f[j] = *reinterpret_cast<const vec*>(src_b + (size_t)n * C + coff[j]);
  };
  // the pixels in walking order: the list segments of a row of cells are contiguous (the list is sorted by cell), so the sequence
  // is row by row [start of the row's first cell, end of its last)
  auto row_range = [&](int r, int& rs, int& re) {
    int s2, L2, Lx;
    cell_range(r * ncx, rs, Lx);
    cell_range(r * ncx + ncx - 1, s2, L2);
    re = s2 + L2;
  };
  struct Seq {
    int i, r, re;     // list index (-1: done), its row of cells, the row's end
  };
  auto seq_from = [&](int r0) -> Seq {
    for (int r = r0; r < ncy; ++r) {
      int rs, re;
      row_range(r, rs, re);
      if (rs < re) return Seq{rs, r, re};
    }
    return Seq{-1, ncy, 0};
  };
  auto seq_next = [&](const Seq& q) -> Seq {
    if (q.i < 0) return q;
    if (q.i + 1 < q.re) return Seq{q.i + 1, q.r, q.re};
    return seq_from(q.r + 1);
  };
  Seq p0 = seq_from(0);
  if (p0.i >= 0) {
    vec tex[4][4][CJ], texn[4][4][CJ], f1[CJ], f1n[CJ];
    f32x4 ra0, ra1, rb0 = {0.f, 0.f, 0.f, 0.f}, rb1 = rb0, rc0 = rb0, rc1 = rb0;     // records of this pixel, the next, the one after
    int cur = next_cell(), s, L;
    cell_range(cur, s, L);
    int e = s + L;
    load_rec(p0.i, ra0, ra1);
    Seq p1 = seq_next(p0);
    if (p1.i >= 0) load_rec(p1.i, rb0, rb1);
    load_tex(cur, tex);
    load_src(__builtin_amdgcn_readfirstlane(__float_as_int(ra0[0])), f1);
    int nxt = next_cell(), sn = 0, Ln = 0;        // the following non-empty cell: its texels travel while this cell's pixels are worked on
    bool want_texn = false;
    if (nxt >= 0) {
      cell_range(nxt, sn, Ln);
      want_texn = true;
    }
    wait_vm0();
    while (true) {
      const int n1 = __builtin_amdgcn_readfirstlane(__float_as_int(rb0[0]));
      const Seq p2 = seq_next(p1);
      if (p2.i >= 0) load_rec(p2.i, rc0, rc1);
      if (p1.i >= 0) load_src(n1, f1n);
      if (want_texn) {
        load_tex(nxt, texn);
        want_texn = false;
      }
      // ---- this pixel: cell (x0, y0), fractions, the per-pixel scalars of adj_pixel*_kernel
      const int r0 = cur / ncx, x0 = cxa + cur - r0 * ncx, y0 = cya + r0;
      const float ax = ra0[1], ay = ra0[2], dg1 = ra0[3], dg2 = ra1[0], dM11 = ra1[1], dM12 = ra1[2], dM22 = ra1[3];
      float cf[4], cgx[4], cgy[4];      // tap weight x (in image, gx defined, gy defined): adj_pixel*_kernel's wt, fin, hx, hy
      vec Sf[CJ], Sgx[CJ], Sgy[CJ];
      if (x0 >= 1 && y0 >= 1 && x0 + 2 <= W - 1 && y0 + 2 <= H - 1) {      // the whole footprint inside the image, off the rim
        const float bx = 1.f - ax, by = 1.f - ay;
        cf[0] = bx * by, cf[1] = ax * by, cf[2] = bx * ay, cf[3] = ax * ay;
#pragma unroll
        for (int t = 0; t < 4; ++t) cgx[t] = cgy[t] = 0.5f * cf[t];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
          Sf[j] = cf[0] * tex[1][1][j] + cf[1] * tex[1][2][j] + cf[2] * tex[2][1][j] + cf[3] * tex[2][2][j];
          Sgx[j] = cgx[0] * (tex[1][2][j] - tex[1][0][j]) + cgx[1] * (tex[1][3][j] - tex[1][1][j]) + cgx[2] * (tex[2][2][j] - tex[2][0][j]) +
                   cgx[3] * (tex[2][3][j] - tex[2][1][j]);
          Sgy[j] = cgy[0] * (tex[2][1][j] - tex[0][1][j]) + cgy[1] * (tex[2][2][j] - tex[0][2][j]) + cgy[2] * (tex[3][1][j] - tex[1][1][j]) +
                   cgy[3] * (tex[3][2][j] - tex[1][2][j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < CJ; ++j) Sf[j] = Sgx[j] = Sgy[j] = vsplat<V>(0.f);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ix = t & 1, iy = t >> 1;
          const int tx = x0 + ix, ty = y0 + iy;
          const bool in = tx <= W - 1 && ty <= H - 1;
          const float hx = (in && tx > 0 && tx < W - 1) ? 0.5f : 0.f, hy = (in && ty > 0 && ty < H - 1) ? 0.5f : 0.f;
          const float fin = in ? 1.f : 0.f;
          const float wx = ix ? ax : 1.f - ax, wy = iy ? ay : 1.f - ay;
          const float wt = wx * wy;
          cf[t] = wt * fin;
          cgx[t] = wt * hx;
          cgy[t] = wt * hy;
#pragma unroll
          for (int j = 0; j < CJ; ++j) {
            Sf[j] += cf[t] * tex[1 + iy][1 + ix][j];
            Sgx[j] += cgx[t] * (tex[1 + iy][2 + ix][j] - tex[1 + iy][ix][j]);
            Sgy[j] += cgy[t] * (tex[2 + iy][1 + ix][j] - tex[iy][1 + ix][j]);
          }
        }
      }
      vec af[CJ], agx[CJ], agy[CJ];     // the adjoint of the sampled (f, gx, gy): what the 3C row held
#pragma unroll
      for (int j = 0; j < CJ; ++j) {
        if (!cok[j]) {
          Sf[j] = Sgx[j] = Sgy[j] = vsplat<V>(0.f);
          f1[j] = vsplat<V>(0.f);
        }
        const vec d = f1[j] - Sf[j];
        vec sg;
        if constexpr (V == 1) {
          sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else {
#pragma unroll
          for (int q = 0; q < V; ++q) sg[q] = d[q] > 0.f ? 1.f : (d[q] < 0.f ? -1.f : 0.f);
        }
        af[j] = -(dg1 * Sgx[j] + dg2 * Sgy[j] + sg * ga[j]);
        agx[j] = 2.f * (dM11 * Sgx[j] + dM12 * Sgy[j]) + dg1 * d;
        agy[j] = 2.f * (dM12 * Sgx[j] + dM22 * Sgy[j]) + dg2 * d;
      }
      // ---- grad_fixed^T into the tile: the 4x4-minus-corners footprint; a texel outside the tile goes to the dummy slot.
      // Plain read-modify-write, all 12 reads before the 12 writes: the accumulators belong to this wave alone (LDS float
      // atomics -- ds_add_f32 -- measured ~240 ns each here: 5.8 us per pixel with 24 of them)
      {
        const int ux = x0 - 1 - tx0, uy = y0 - 1 - ty0;     // footprint column cc / row r -> tile column ux + cc / row uy + r
        int xo[4], yo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xo[k] = (unsigned)(ux + k) < (unsigned)TW ? (ux + k) * kSlotV : -1;
          yo[k] = (unsigned)(uy + k) < (unsigned)TH ? (uy + k) * TW * kSlotV : -1;
        }
        constexpr int kR[12] = {1, 1, 2, 2, 0, 3, 0, 3, 1, 1, 2, 2}, kC[12] = {0, 3, 0, 3, 1, 1, 2, 2, 1, 2, 1, 2};
        vec* dst[12];
        vec val[12][CJ], old[12][CJ];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const int o = (xo[kC[k]] | yo[kR[k]]) < 0 ? TW * TH * kSlotV : xo[kC[k]] + yo[kR[k]];
          dst[k] = sAccV + o + lane;
        }
        // gx of tap (ix, iy) = hx (T[tx + 1] - T[tx - 1]), gy = hy (T[ty + 1] - T[ty - 1]); an inner texel also takes its tap's own
        // value, the x-neighbour tap's gx (it is that tap's right / left neighbour) and the y-neighbour tap's gy
        const float k8[8] = {-cgx[0], cgx[1], -cgx[2], cgx[3], -cgy[0], cgy[2], -cgy[1], cgy[3]};
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
#pragma unroll
          for (int k = 0; k < 4; ++k) val[k][j] = k8[k] * agx[j];
#pragma unroll
          for (int k = 4; k < 8; ++k) val[k][j] = k8[k] * agy[j];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int ix = t & 1, iy = t >> 1;
            const float kx = ix ? cgx[2 * iy] : -cgx[2 * iy + 1], ky = iy ? cgy[ix] : -cgy[2 + ix];
            val[8 + t][j] = cf[t] * af[j] + kx * agx[j] + ky * agy[j];
          }
        }
#pragma unroll
        for (int k = 0; k < 12; ++k)
#pragma unroll
          for (int j = 0; j < CJ; ++j) old[k][j] = dst[k][j * 64];
#pragma unroll
        for (int k = 0; k < 12; ++k)
#pragma unroll
          for (int j = 0; j < CJ; ++j) dst[k][j * 64] = old[k][j] + val[k][j];
      }
      // ---- advance: everything requested above has had the arithmetic to arrive
      if (p1.i < 0) break;
      wait_vm0();
      ra0 = rb0;
      ra1 = rb1;
      rb0 = rc0;
      rb1 = rc1;
#pragma unroll
      for (int j = 0; j < CJ; ++j) f1[j] = f1n[j];
      if (p1.i >= e) {                  // into the next non-empty cell: its texels become current, the cell after it is requested
        cur = nxt;
        e = sn + Ln;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int j = 0; j < CJ; ++j) tex[r][cc][j] = texn[r][cc][j];
        nxt = next_cell();
        if (nxt >= 0) {
          cell_range(nxt, sn, Ln);
          want_texn = true;             // requested at the top of the next iteration
        }
      }
      p0 = p1;
      p1 = p2;
    }
  }
  // ---- the tile's texels: written once (overwrite_map) or added to the gradient of the earlier iterations
  float* __restrict__ out_b = a.dmap3 + (size_t)b * HW * C;
  for (int t = 0; t < TW * TH; ++t) {
    const int X = tx0 + (t % TW), Y = ty0 + t / TW;
    if (X >= W || Y >= H) continue;
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      if (!cok[j]) continue;
      const vec v = sAccV[(t * CJ + j) * 64 + lane];
      vec* o = reinterpret_cast<vec*>(out_b + (size_t)(Y * W + X) * C + coff[j]);
      *o = a.overwrite_map ? v : *o + v;
    }
  }
}

// ---- two visits per wave instruction (C % 4 == 0, C <= 128): a HALF wave per pixel, 4 channels per lane --------------------------
// adj_tile_kernel above is bound by VALU issue: ~450 instructions per visit of which only a third is channel arithmetic (measured:
// no tile shape moves it, 1420 SIMD cycles per visit).  Here lanes 0-31 and 32-63 work on two different pixels of the tile's walking
// order (visit k and visit k + half: the two halves of the sequence, so that the pair's footprints are usually disjoint), every
// per-pixel quantity is a per-lane value, every row access one 16-byte load per lane: the same instruction stream serves two visits.
//   * walking order = the cell rows' list segments back to back; a lane finds its list index by a select chain over the row table
//     (<= TH + 3 rows, uniform); lidx gives (pixel, cell), lrec the fractions and the five scalars;
//   * two register sets (this pair / next pair), the loop unrolled twice so that nothing is copied; the (pixel, cell) pair two
//     iterations ahead; one full s_waitcnt vmcnt(0) per iteration after the arithmetic (see adj_tile_kernel);
//   * accumulators: (TH + 1) x (TW + 1) slots of 32 lanes x 16 bytes -- the extra row and column take what falls outside the tile,
//     so a destination's address is row offset + column offset (unsigned min clamps either to the dummy), no select per texel;
//   * the two halves' 12-texel footprints may overlap: then the read-modify-write runs half by half (two exec-masked phases, in
//     order: deterministic), else once for the whole wave.
typedef float f32x3 __attribute__((ext_vector_type(3)));
template <int TW, int TH>
__global__ __launch_bounds__(64) void adj_tile2_kernel(const AdjArgs a, int tiles_x, int ntiles, int total, int chunk) {
  constexpr int MAXR = TH + 3, SW = TW + 1, SH = TH + 1;
  extern __shared__ __attribute__((aligned(16))) float sAcc[];
  f32x4* sAccV = reinterpret_cast<f32x4*>(sAcc);               // slot (uy, ux), lane hl: (uy SW + ux) 32 + hl
  const int lane = threadIdx.x, hl = lane & 31, hi = lane >> 5;
  const int kk = blockIdx.x >> 3, work = (blockIdx.x & 7) * chunk + kk;
  if (kk >= chunk || work >= total) return;
  const int b = work / ntiles, tile = work - b * ntiles;
  const int tyi = tile / tiles_x, tx0 = (tile - tyi * tiles_x) * TW, ty0 = tyi * TH;
  const int N = a.lv.N, C = a.lv.C, H = a.lv.H, W = a.lv.W, HW = H * W;
  const float* __restrict__ src_b = a.lv.src + (size_t)b * N * C;
  const float* __restrict__ tgt_b = a.lv.tgt + (size_t)b * HW * C;
  const float* __restrict__ lrec = a.lrec + (size_t)b * N * kFrac;
  const int2* __restrict__ lidx = a.lidx + (size_t)b * N;
  const int2* __restrict__ cs = reinterpret_cast<const int2*>(a.start) + (size_t)b * HW;
  const bool cok = 4 * hl < C;
  const int coff = cok ? 4 * hl : 0;
  f32x4 ga = *reinterpret_cast<const f32x4*>(a.gabs + (size_t)b * C + coff);
  if (!cok) ga = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < SH * SW * 32; i += 64) sAccV[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the rows of cells whose pixels can touch the tile, their list segments, the running visit count (all uniform)
  const int cxa = max(tx0 - 2, 0), cxb = min(tx0 + TW, W - 1);
  const int cya = max(ty0 - 2, 0), cyb = min(ty0 + TH, H - 1), ncy = cyb - cya + 1;
  int rsv = 0, cntv = 0;
  if (lane < ncy) {
    const int2 c0 = cs[(cya + lane) * W + cxa], c1 = cs[(cya + lane) * W + cxb];
    rsv = c0.x;
    cntv = c1.x + c1.y - c0.x;
  }
  // visit v of row r has list index v + base_r, base_r = (start of the row's segment) - (visits before the row); a lane adds up the
  // steps base_r - base_{r-1} of the rows it has passed (a sum of conditional terms: a select CHAIN over base_r is turned into a
  // lookup table in scratch memory by the compiler, one scratch load per index)
  int cum[MAXR + 1], step[MAXR];
  cum[0] = 0;
  int prev = 0;
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int cnt = __builtin_amdgcn_readlane(cntv, r);         // (rows >= ncy hold 0)
    const int bs = __builtin_amdgcn_readlane(rsv, r) - cum[r];
    step[r] = bs - prev;
    prev = bs;
    cum[r + 1] = cum[r] + cnt;
  }
  const int nv = cum[MAXR], half = (nv + 1) >> 1;
  auto list_index = [&](int k) -> int {      // the list index of this lane's visit in iteration k (clamped to a valid one)
    const int v = min(k + hi * half, nv - 1);
    int idx = v + step[0];
#pragma unroll
    for (int r = 1; r < MAXR; ++r) idx += step[r] & -(int)(v >= cum[r]);
    return idx;
  };
  struct Set {                 // one pair of visits: everything its arithmetic needs
    f32x3 r0;                  // (ax, ay, dg1) -- 12 bytes: a 16-byte load would leave a dead register (the pixel index) that the
    f32x4 r1;                  // compiler re-uses at once, waiting for the load it belongs to;  (dg2, dM11, dM12, dM22)
    f32x4 f1;                  // the pixel's source row, this lane's 4 channels
    f32x4 tex[4][4];           // the 4x4-minus-corners target texels of its cell
    int xy;                    // x0 | y0 << 16
  };
  // every row address = a wave-uniform base + an unsigned 32-bit BYTE offset (adj_fold_supported: a window's maps stay below 4 GB):
  // the loads take the base from scalar registers, no 64-bit address arithmetic per load
  const char* const src_c = reinterpret_cast<const char*>(src_b);
  const char* const tgt_c = reinterpret_cast<const char*>(tgt_b);
  const char* const lrec_c = reinterpret_cast<const char*>(lrec);
  const unsigned rowb = (unsigned)C * 4u, coffb = (unsigned)coff * 4u, pitchb = (unsigned)W * rowb;
  auto request = [&](int i, const int2 nx, Set& q) {
    const unsigned ib = (unsigned)i * (unsigned)(kFrac * 4);
    q.r0 = *reinterpret_cast<const f32x3*>(lrec_c + (ib + 4u));
    q.r1 = *reinterpret_cast<const f32x4*>(lrec_c + (ib + 16u));
    q.f1 = *reinterpret_cast<const f32x4*>(src_c + (__umul24((unsigned)nx.x, rowb) + coffb));      // (24-bit multiplies: full rate; N, W C 4 < 2^24: adj_fold_supported)
    q.xy = nx.y;
    const int cx = nx.y & 0xffff, cy = nx.y >> 16;
    unsigned xo[4], yo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xo[k] = __umul24((unsigned)min(max(cx - 1 + k, 0), W - 1), rowb) + coffb;
      yo[k] = __umul24((unsigned)min(max(cy - 1 + k, 0), H - 1), pitchb);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if ((r == 0 || r == 3) && (cc == 0 || cc == 3)) continue;
        q.tex[r][cc] = *reinterpret_cast<const f32x4*>(tgt_c + (yo[r] + xo[cc]));
      }
  };
  auto work_on = [&](int k, const Set& q) {
    const bool live = k + hi * half < nv;                       // (an odd count: the upper half idles in the last iteration)
    const int x0 = q.xy & 0xffff, y0 = q.xy >> 16;
    const float ax = q.r0[0], ay = q.r0[1], dg1 = q.r0[2], dg2 = q.r1[0], dM11 = q.r1[1], dM12 = q.r1[2], dM22 = q.r1[3];
    float cf[4], cgx[4], cgy[4];      // tap weight x (in image, gx defined, gy defined): adj_pixel*_kernel's wt, fin, hx, hy
    const bool inner = x0 >= 1 && y0 >= 1 && x0 + 2 <= W - 1 && y0 + 2 <= H - 1;
    if (__all(inner)) {               // (wave-uniform) both footprints inside the image, off the rim
      const float bx = 1.f - ax, by = 1.f - ay;
      cf[0] = bx * by, cf[1] = ax * by, cf[2] = bx * ay, cf[3] = ax * ay;
#pragma unroll
      for (int t = 0; t < 4; ++t) cgx[t] = cgy[t] = 0.5f * cf[t];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ix = t & 1, iy = t >> 1;
        const int tx = x0 + ix, ty = y0 + iy;
        const bool in = tx <= W - 1 && ty <= H - 1;
        const float hx = (in && tx > 0 && tx < W - 1) ? 0.5f : 0.f, hy = (in && ty > 0 && ty < H - 1) ? 0.5f : 0.f;
        const float wx = ix ? ax : 1.f - ax, wy = iy ? ay : 1.f - ay;
        const float wt = wx * wy;
        cf[t] = in ? wt : 0.f;
        cgx[t] = wt * hx;
        cgy[t] = wt * hy;
      }
    }
    f32x4 Sf = cf[0] * q.tex[1][1] + cf[1] * q.tex[1][2] + cf[2] * q.tex[2][1] + cf[3] * q.tex[2][2];
    f32x4 Sgx = cgx[0] * (q.tex[1][2] - q.tex[1][0]) + cgx[1] * (q.tex[1][3] - q.tex[1][1]) + cgx[2] * (q.tex[2][2] - q.tex[2][0]) +
                cgx[3] * (q.tex[2][3] - q.tex[2][1]);
    f32x4 Sgy = cgy[0] * (q.tex[2][1] - q.tex[0][1]) + cgy[1] * (q.tex[2][2] - q.tex[0][2]) + cgy[2] * (q.tex[3][1] - q.tex[1][1]) +
                cgy[3] * (q.tex[3][2] - q.tex[1][2]);
    f32x4 f1 = q.f1;
    if (!cok || !live) {
      Sf = Sgx = Sgy = f32x4{0.f, 0.f, 0.f, 0.f};
      f1 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 d = f1 - Sf;
    f32x4 sg;
#pragma unroll
    for (int e = 0; e < 4; ++e) sg[e] = d[e] > 0.f ? 1.f : (d[e] < 0.f ? -1.f : 0.f);
    const f32x4 af = -(dg1 * Sgx + dg2 * Sgy + sg * ga);
    const f32x4 agx = 2.f * (dM11 * Sgx + dM12 * Sgy) + dg1 * d;
    const f32x4 agy = 2.f * (dM12 * Sgx + dM22 * Sgy) + dg2 * d;
    // ---- grad_fixed^T into the tile (see adj_tile_kernel); footprint column cc / row r -> slot column ux + cc / row uy + r, clamped
    // (unsigned) to the dummy column TW / row TH; an idle half goes to the dummy corner altogether
    const int ux = live ? x0 - 1 - tx0 : TW, uy = live ? y0 - 1 - ty0 : TH;
    unsigned xo[4], yo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xo[k] = min((unsigned)(ux + k), (unsigned)TW) * 32u + (unsigned)hl;
      yo[k] = min((unsigned)(uy + k), (unsigned)TH) * (unsigned)(SW * 32);
    }
    constexpr int kR[12] = {1, 1, 2, 2, 0, 3, 0, 3, 1, 1, 2, 2}, kC[12] = {0, 3, 0, 3, 1, 1, 2, 2, 1, 2, 1, 2};
    const float k8[8] = {-cgx[0], cgx[1], -cgx[2], cgx[3], -cgy[0], cgy[2], -cgy[1], cgy[3]};
    // do the two halves' footprints overlap?  (uniform: the other half's cell through a lane read)
    const int xyo = __builtin_amdgcn_readlane(q.xy, 32), xyl = __builtin_amdgcn_readlane(q.xy, 0);
    const int ddx = (xyo & 0xffff) - (xyl & 0xffff), ddy = (xyo >> 16) - (xyl >> 16);
    const bool apart = ddx >= 4 || ddx <= -4 || ddy >= 4 || ddy <= -4 || k + half >= nv;
    {                                     // all 12 destinations at once: 48 registers of values + 48 of old contents
      constexpr int ND = 12;
      f32x4 val[ND];
      f32x4* dst[ND];
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int kd = u;
        dst[u] = sAccV + (yo[kR[kd]] + xo[kC[kd]]);
        if (kd < 4) {
          val[u] = k8[kd] * agx;
        } else if (kd < 8) {
          val[u] = k8[kd] * agy;
        } else {
          const int t = kd - 8, ix = t & 1, iy = t >> 1;
          const float kx = ix ? cgx[2 * iy] : -cgx[2 * iy + 1], ky = iy ? cgy[ix] : -cgy[2 + ix];
          val[u] = cf[t] * af + kx * agx + ky * agy;
        }
      }
      if (apart) {
        f32x4 old[ND];
#pragma unroll
        for (int u = 0; u < ND; ++u) old[u] = *dst[u];
#pragma unroll
        for (int u = 0; u < ND; ++u) *dst[u] = old[u] + val[u];
      } else {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          if (hi == ph) {
            f32x4 old[ND];
#pragma unroll
            for (int u = 0; u < ND; ++u) old[u] = *dst[u];
#pragma unroll
            for (int u = 0; u < ND; ++u) *dst[u] = old[u] + val[u];
          }
          // the other half reads what this half wrote: lanes of one wave, but different THREADS to the compiler -- without the
          // fence it may move the second phase's loads above the first phase's (exec-masked) stores
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (and the next visit's reads stay behind this visit's writes)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  if (nv > 0) {
    Set A, B;
    int2 nx1, nx2 = make_int2(0, 0);
    int i1, i2;                           // the list indices of iterations k + 1 / k + 2 (each computed once: ~20 instructions)
    {
      const int i0 = list_index(0);
      const int2 nx0 = lidx[i0];
      request(i0, nx0, A);
      i1 = list_index(1);                 // (clamped: valid even when there is no second iteration)
      nx1 = lidx[i1];
    }
    wait_vm0();
    for (int k = 0; k < half; k += 2) {
      // iteration k on A: request k + 1 into B, the (pixel, cell) of k + 2.  Unconditionally (list_index clamps past the end: the
      // last iteration requests a valid pair it never uses) -- a conditional request makes the compiler merge the register set with
      // its old contents by copies that wait for the loads just issued
      request(i1, nx1, B);
      i2 = list_index(k + 2);
      nx2 = lidx[i2];
      work_on(k, A);
      wait_vm0();
      nx1 = nx2;
      i1 = i2;
      if (k + 1 >= half) break;
      // iteration k + 1 on B: request k + 2 into A
      request(i1, nx1, A);
      i2 = list_index(k + 3);
      nx2 = lidx[i2];
      work_on(k + 1, B);
      wait_vm0();
      nx1 = nx2;
      i1 = i2;
    }
  }
  // ---- the tile's texels: written once (overwrite_map) or added to the gradient of the earlier iterations; a half wave per texel
  float* __restrict__ out_b = a.dmap3 + (size_t)b * HW * C;
  for (int t2 = 0; t2 < TW * TH; t2 += 2) {
    const int t = t2 + hi;
    const int uy = t / TW, ux = t - uy * TW, X = tx0 + ux, Y = ty0 + uy;
    if (t >= TW * TH || X >= W || Y >= H || !cok) continue;
    const f32x4 v = sAccV[(uy * SW + ux) * 32 + hl];
    f32x4* o = reinterpret_cast<f32x4*>(out_b + (size_t)(Y * W + X) * C + coff);
    *o = a.overwrite_map ? v : *o + v;
  }
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s, 64));
  return v;
}

// One wave per target texel: the adjoint of the bilinear sampling as a GATHER over the source pixels whose footprint
// covers the texel, in a fixed order (cells row-major, ascending pixel index inside a cell).  Common case (every cell
// holds at most one pixel, e.g. near-unit local scale): straight-line code, 4 + 4 + 4 independent loads, then the rows.
template <int J3>   // 3C <= 64 J3; one texel, the whole wave
__device__ __forceinline__ void adj_map_texel(const AdjArgs& a, int b, int t, int lane) {
  const int N = a.lv.N, C = a.lv.C, H = a.lv.H, W = a.lv.W, HW = H * W, C3 = 3 * C;
  const int2* __restrict__ cs = reinterpret_cast<const int2*>(a.start) + (size_t)b * HW;
  const int* __restrict__ list = a.list + (size_t)b * N;
  const float* __restrict__ frac = a.frac + (size_t)b * N * kFrac;
  const float* __restrict__ arow = a.arow + (size_t)b * N * C3;
  bool cok[J3];
#pragma unroll
  for (int j = 0; j < J3; ++j) cok[j] = lane + 64 * j < C3;
  {
    const int ty = t / W, tx = t - ty * W;
    int L[4], s0[4];
#pragma unroll
    for (int cell = 0; cell < 4; ++cell) {
      const int cy = ty - 1 + (cell >> 1), cx = tx - 1 + (cell & 1);
      const bool okc = cy >= 0 && cx >= 0;
      const int2 v = cs[okc ? cy * W + cx : 0];
      L[cell] = okc ? v.y : 0;
      s0[cell] = v.x;
    }
    const int Lmax = max(max(L[0], L[1]), max(L[2], L[3]));
    if (Lmax == 0) {         // wave-uniform: nothing lands on this texel
      if (a.overwrite_map) {
        float* __restrict__ o = a.dmap3 + ((size_t)b * HW + t) * C3;
#pragma unroll
        for (int j = 0; j < J3; ++j)
          if (cok[j]) o[lane + 64 * j] = 0.f;
      }
      return;
    }
    float acc[J3];
#pragma unroll
    for (int j = 0; j < J3; ++j) acc[j] = 0.f;
    bool any = false;
    if (Lmax == 1) {
      int n1[4];
#pragma unroll
      for (int cell = 0; cell < 4; ++cell) n1[cell] = list[L[cell] ? s0[cell] : 0];
      float wt[4];
#pragma unroll
      for (int cell = 0; cell < 4; ++cell) {
        const float fax = frac[(size_t)n1[cell] * kFrac + 1], fay = frac[(size_t)n1[cell] * kFrac + 2];
        const float v = ((cell & 1) ? 1.f - fax : fax) * ((cell >> 1) ? 1.f - fay : fay);   // cell = texel - (1,1) ... texel
        wt[cell] = L[cell] ? v : 0.f;
        any = any || wt[cell] != 0.f;
      }
      float val[4][J3];
#pragma unroll
      for (int cell = 0; cell < 4; ++cell)
#pragma unroll
        for (int j = 0; j < J3; ++j) val[cell][j] = arow[(size_t)n1[cell] * C3 + (cok[j] ? lane + 64 * j : 0)];
#pragma unroll
      for (int cell = 0; cell < 4; ++cell)
#pragma unroll
        for (int j = 0; j < J3; ++j) acc[j] = fmaf(wt[cell], wt[cell] != 0.f ? val[cell][j] : 0.f, acc[j]);
    } else {
#pragma unroll 1
      for (int cell = 0; cell < 4; ++cell) {
        int last = -1;
        for (int it = 0; it < L[cell]; ++it) {
          int mn = 0x7fffffff;
          for (int e = lane; e < L[cell]; e += 64) {
            const int v = list[s0[cell] + e];
            if (v > last) mn = min(mn, v);
          }
          const int n = wave_min_i(mn);
          last = n;
          const float ax = frac[(size_t)n * kFrac + 1], ay = frac[(size_t)n * kFrac + 2];
          const float wt = ((cell & 1) ? 1.f - ax : ax) * ((cell >> 1) ? 1.f - ay : ay);
          if (wt != 0.f) {
            any = true;
#pragma unroll
            for (int j = 0; j < J3; ++j) acc[j] = fmaf(wt, arow[(size_t)n * C3 + (cok[j] ? lane + 64 * j : 0)], acc[j]);
          }
        }
      }
    }
    if (any || a.overwrite_map) {   // (acc is all zeros when nothing contributed)
      float* __restrict__ o = a.dmap3 + ((size_t)b * HW + t) * C3;
#pragma unroll
      for (int j = 0; j < J3; ++j)
        if (cok[j]) o[lane + 64 * j] = a.overwrite_map ? acc[j] : o[lane + 64 * j] + acc[j];
    }
  }
}

template <int J3>
__global__ __launch_bounds__(kBlock) void adj_map_kernel(const AdjArgs a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = wave_id();
  const int HW = a.lv.H * a.lv.W;
  for (int t = blockIdx.x * kNumWaves + w; t < HW; t += gridDim.x * kNumWaves) adj_map_texel<J3>(a, b, t, lane);
}

// Two texels per wave (a half wave each, 16-byte accesses) for the common case that every cell involved holds at most one
// pixel; a pair with a fuller cell goes through adj_map_texel, one texel after the other.  Same fma sequence per element as
// adj_map_kernel: identical bits.  3C % 4 == 0, 3C <= 128 J4.
template <int J4, int J3>
__global__ __launch_bounds__(kBlock) void adj_map2_kernel(const AdjArgs a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = wave_id(), hl = lane & 31, hi = lane >> 5;
  const int N = a.lv.N, C = a.lv.C, H = a.lv.H, W = a.lv.W, HW = H * W, C3 = 3 * C;
  const int2* __restrict__ cs = reinterpret_cast<const int2*>(a.start) + (size_t)b * HW;
  const int* __restrict__ list = a.list + (size_t)b * N;
  const float* __restrict__ frac = a.frac + (size_t)b * N * kFrac;
  const float* __restrict__ arow = a.arow + (size_t)b * N * C3;
  bool cok[J4];
#pragma unroll
  for (int j = 0; j < J4; ++j) cok[j] = 4 * hl + 128 * j < C3;
  for (int t2 = 2 * (blockIdx.x * kNumWaves + w); t2 < HW; t2 += 2 * gridDim.x * kNumWaves) {
    const int t = t2 + hi;
    const bool live = t < HW;
    const int tt = live ? t : HW - 1;
    const int ty = tt / W, tx = tt - ty * W;
    int L[4], s0[4];
#pragma unroll
    for (int cell = 0; cell < 4; ++cell) {
      const int cy = ty - 1 + (cell >> 1), cx = tx - 1 + (cell & 1);
      const bool okc = live && cy >= 0 && cx >= 0;
      const int2 v = cs[okc ? cy * W + cx : 0];
      L[cell] = okc ? v.y : 0;
      s0[cell] = v.x;
    }
    const int Lmax = max(max(L[0], L[1]), max(L[2], L[3]));
    if (__any(Lmax > 1)) {                       // wave-uniform: a fuller cell somewhere in the pair
      adj_map_texel<J3>(a, b, t2, lane);
      if (t2 + 1 < HW) adj_map_texel<J3>(a, b, t2 + 1, lane);
      continue;
    }
    if (!__any(Lmax == 1)) {                     // nothing lands on either texel
      if (a.overwrite_map && live) {
        float* __restrict__ o = a.dmap3 + ((size_t)b * HW + t) * C3 + 4 * hl;
#pragma unroll
        for (int j = 0; j < J4; ++j)
          if (cok[j]) *reinterpret_cast<f32x4*>(o + 128 * j) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      continue;
    }
    int n1[4];
#pragma unroll
    for (int cell = 0; cell < 4; ++cell) n1[cell] = list[L[cell] ? s0[cell] : 0];
    float wt[4];
    bool any = false;
#pragma unroll
    for (int cell = 0; cell < 4; ++cell) {
      const float fax = frac[(size_t)n1[cell] * kFrac + 1], fay = frac[(size_t)n1[cell] * kFrac + 2];
      const float v = ((cell & 1) ? 1.f - fax : fax) * ((cell >> 1) ? 1.f - fay : fay);
      wt[cell] = L[cell] ? v : 0.f;
      any = any || wt[cell] != 0.f;
    }
    f32x4 val[4][J4];
#pragma unroll
    for (int cell = 0; cell < 4; ++cell)
#pragma unroll
      for (int j = 0; j < J4; ++j) val[cell][j] = *reinterpret_cast<const f32x4*>(arow + (size_t)n1[cell] * C3 + (cok[j] ? 4 * hl + 128 * j : 0));
    f32x4 acc[J4];
#pragma unroll
    for (int j = 0; j < J4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cell = 0; cell < 4; ++cell)
#pragma unroll
      for (int j = 0; j < J4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(wt[cell], wt[cell] != 0.f ? val[cell][j][e] : 0.f, acc[j][e]);
    if (any || (a.overwrite_map && live)) {           // (acc is all zeros when nothing contributed)
      float* __restrict__ o = a.dmap3 + ((size_t)b * HW + t) * C3 + 4 * hl;
#pragma unroll
      for (int j = 0; j < J4; ++j)
        if (cok[j]) {
          f32x4* op = reinterpret_cast<f32x4*>(o + 128 * j);
          *op = a.overwrite_map ? acc[j] : *op + acc[j];
        }
    }
  }
}

// d img += dmap_f + grad_fixed^T (dmap_gx, dmap_gy): gx[x] = 0.5 (img[x+1] - img[x-1]) for 1 <= x <= W-2, 0 on the rim
__global__ void target_map_adjoint_kernel(const float* __restrict__ dmap3, float* __restrict__ dimg, int H, int W, int C,
                                          size_t total, int overwrite) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const size_t tix = e / C;
  const int x = (int)(tix % W);
  const size_t r = tix / W;
  const int y = (int)(r % H);
  const size_t C3 = 3 * (size_t)C;
  const float* m = dmap3 + tix * C3;
  float v = m[c];
  if (x - 1 >= 1 && x - 1 <= W - 2) v += 0.5f * (m - C3)[C + c];
  if (x + 1 >= 1 && x + 1 <= W - 2) v -= 0.5f * (m + C3)[C + c];
  if (y - 1 >= 1 && y - 1 <= H - 2) v += 0.5f * (m - (size_t)W * C3)[2 * C + c];
  if (y + 1 >= 1 && y + 1 <= H - 2) v -= 0.5f * (m + (size_t)W * C3)[2 * C + c];
  dimg[e] = overwrite ? v : dimg[e] + v;
}

// C % 4 == 0: four channels per thread, 16-byte accesses; per element the same operations in the same order (identical bits)
__global__ void target_map_adjoint4_kernel(const float* __restrict__ dmap3, float* __restrict__ dimg, int H, int W, int C4,
                                           size_t total4, int overwrite) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total4) return;
  const int c = (int)(e % C4);
  const size_t tix = e / C4;
  const int x = (int)(tix % W);
  const size_t r = tix / W;
  const int y = (int)(r % H);
  const size_t Q3 = 3 * (size_t)C4;                      // float4 per texel of the [f|gx|gy] map
  const float4* m = reinterpret_cast<const float4*>(dmap3) + tix * Q3;
  float4 v = m[c];
  auto axpy = [&](float a, const float4 t) {
    v.x += a * t.x;
    v.y += a * t.y;
    v.z += a * t.z;
    v.w += a * t.w;
  };
  if (x - 1 >= 1 && x - 1 <= W - 2) axpy(0.5f, (m - Q3)[C4 + c]);
  if (x + 1 >= 1 && x + 1 <= W - 2) axpy(-0.5f, (m + Q3)[C4 + c]);
  if (y - 1 >= 1 && y - 1 <= H - 2) axpy(0.5f, (m - (size_t)W * Q3)[2 * C4 + c]);
  if (y + 1 >= 1 && y + 1 <= H - 2) axpy(-0.5f, (m + (size_t)W * Q3)[2 * C4 + c]);
  float4* o = reinterpret_cast<float4*>(dimg) + e;
  if (overwrite) {
    *o = v;
    return;
  }
  float4 d = *o;
  d.x += v.x;
  d.y += v.y;
  d.z += v.z;
  d.w += v.w;
  *o = d;
}

// ---- the same deterministic gather for the reference-layout adjoint (sparse points, [f|gx|gy] target map) ----------
// ba_sample_stats_grad_kernel (sstats.hip) scatters its 3C-wide contributions with float atomics; this variant writes them as
// per-point rows + (cell, fractions) records and lets adj_scan / adj_fill / adj_map_kernel gather them per texel in a fixed
// order: bit-reproducible dconv2.  One wave per point at a time, lane = channel (+64 j).
__global__ __launch_bounds__(kBlock) void sstats_rows_kernel(const float* __restrict__ conv1, const float* __restrict__ conv2,
                                                             const float* __restrict__ px, const float* __restrict__ py, int N,
                                                             int C, int H, int W, const float* __restrict__ dstats,
                                                             const float* __restrict__ dabs, float* __restrict__ dconv1,
                                                             float* __restrict__ dpos, float* __restrict__ arow,
                                                             float* __restrict__ frac, int* __restrict__ cnt) {
  constexpr int kPix = 16;
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, w = wave_id();
  const int CJ = (C + 63) >> 6, C3 = 3 * C;
  float da[kAdjMaxCJ];
  for (int j = 0; j < kAdjMaxCJ; ++j) {
    const int c = lane + 64 * j;
    da[j] = c < C ? dabs[(size_t)b * C + c] : 0.f;
  }
  for (int i = 0; i < kPix; ++i) {
    const int n = g * kPix * kNumWaves + w * kPix + i;
    if (n >= N) break;                                   // wave-uniform
    const size_t q = (size_t)b * N + n;
    const float pxv = px[q], pyv = py[q];
    const bool m = (pxv >= 0.f) && (pxv <= (float)(W - 1)) && (pyv >= 0.f) && (pyv <= (float)(H - 1));   // bundlenet.py:231-233
    float dpx = 0.f, dpy = 0.f;
    if (!m) {
      for (int j = 0; j < CJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C) dconv1[q * C + c] = 0.f;
      }
      if (lane == 0) frac[q * kFrac] = __int_as_float(-1);
    } else {
      const float xf = floorf(pxv), yf = floorf(pyv);
      const int x0 = (int)xf, y0 = (int)yf;
      const float ax = pxv - xf, ay = pyv - yf;
      const bool xi = x0 + 1 <= W - 1, yi = y0 + 1 <= H - 1;
      const int xc = xi ? x0 + 1 : x0, yc = yi ? y0 + 1 : y0;
      const float* __restrict__ base = conv2 + (size_t)b * H * W * C3;
      const float* __restrict__ r0 = base + (size_t)(y0 * W + x0) * C3;
      const float* __restrict__ r1 = base + (size_t)(y0 * W + xc) * C3;
      const float* __restrict__ r2 = base + (size_t)(yc * W + x0) * C3;
      const float* __restrict__ r3 = base + (size_t)(yc * W + xc) * C3;
      const float w0 = (1.f - ax) * (1.f - ay), w1 = xi ? ax * (1.f - ay) : 0.f, w2 = yi ? (1.f - ax) * ay : 0.f,
                  w3 = (xi && yi) ? ax * ay : 0.f;
      const float4 s0 = *reinterpret_cast<const float4*>(dstats + q * 8);
      const float dm11 = s0.x, dm12 = s0.y, dm22 = s0.z, dg1 = s0.w, dg2 = dstats[q * 8 + 4];
      for (int j = 0; j < CJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
          float v[4][3];
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            v[0][e] = r0[e * C + c];
            v[1][e] = xi ? r1[e * C + c] : 0.f;
            v[2][e] = yi ? r2[e * C + c] : 0.f;
            v[3][e] = (xi && yi) ? r3[e * C + c] : 0.f;
          }
          float s[3];
#pragma unroll
          for (int e = 0; e < 3; ++e) s[e] = w0 * v[0][e] + w1 * v[1][e] + w2 * v[2][e] + w3 * v[3][e];
          const float gx = s[1], gy = s[2], d = conv1[q * C + c] - s[0];
          const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
          const float dd = gx * dg1 + gy * dg2 + sgn * da[j];
          float dv[3];
          dv[0] = -dd;                                              // d = conv1 - f2
          dv[1] = 2.f * gx * dm11 + gy * dm12 + d * dg1;
          dv[2] = 2.f * gy * dm22 + gx * dm12 + d * dg2;
          dconv1[q * C + c] = dd;
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            arow[q * C3 + e * C + c] = dv[e];
            dpx = fmaf(dv[e], (1.f - ay) * (v[1][e] - v[0][e]) + ay * (v[3][e] - v[2][e]), dpx);
            dpy = fmaf(dv[e], (1.f - ax) * (v[2][e] - v[0][e]) + ax * (v[3][e] - v[1][e]), dpy);
          }
        }
      }
      dpx = wave_sum(dpx);
      dpy = wave_sum(dpy);
      if (lane == 0) {
        const int key = y0 * W + x0;
        frac[q * kFrac] = __int_as_float(key);
        frac[q * kFrac + 1] = ax;
        frac[q * kFrac + 2] = ay;
        atomicAdd(&cnt[(size_t)b * H * W + key], 1);
      }
    }
    if (lane == 0) *reinterpret_cast<float2*>(dpos + q * 2) = make_float2(dpx, dpy);
  }
}

static void launch_adj_map(const AdjArgs& a, int C, dim3 grid, dim3 block, hipStream_t s) {
  const int C3 = 3 * C, J3 = (C3 + 63) / 64;
  const bool half = (C3 & 3) == 0 && !(a.lv.flags & (1 << 28));   // bit 28: one texel per wave (A/B; bits 18 / 19 belong to the forward's gather selection)
  if (J3 <= 3) {
    if (half)
      hipLaunchKernelGGL((adj_map2_kernel<2, 3>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((adj_map_kernel<3>), grid, block, 0, s, a);
  } else if (J3 <= 6) {
    if (half)
      hipLaunchKernelGGL((adj_map2_kernel<3, 6>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((adj_map_kernel<6>), grid, block, 0, s, a);
  } else {
    if (half)
      hipLaunchKernelGGL((adj_map2_kernel<6, 12>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((adj_map_kernel<12>), grid, block, 0, s, a);
  }
}

struct AdjPlan {
  int G, Ga, Gm;
  int fold;           // adj_tile_kernel instead of the 3C rows + per-texel gather (BANET_ADJOINT_FOLD_TARGET)
  int bigq_cap;
  size_t off_S, off_z2, off_arec, off_arow, off_frac, off_cnt, off_start, off_cursor, off_list, off_part, off_chunks, off_lrec, off_list2,
      off_lidx, bytes;
};

bool adj_supported(const banet_level_t* lv) {
  const bool var_ok = (lv->variant == BANET_BUNDLE && lv->K >= 1 && lv->K <= 64 * kAdjMaxKJ) ||
                      (lv->variant == BANET_BUNDLE_CAMERA && lv->K == 0);     // pose only: bundlenet.py:122-191, depth fixed
  const bool dense_ok = lv->dense == 1 && lv->tgt_has_grad == 0 && lv->N == lv->H * lv->W;
  // round 5: sparse points in the reference's layout (conv1 [B,N,C], rays + per-point intrinsics, [f|gx|gy] target map)
  const bool sparse_ok = lv->dense == 0 && lv->tgt_has_grad == 1 && lv->rays && lv->fx && lv->fy && lv->ox && lv->oy && lv->H >= 2 && lv->W >= 2 &&
                         !(lv->C > 128 && lv->K > 128);     // (the point kernel is not compiled for > 2 channel chunks with > 2 coefficient chunks)
  return var_ok && (dense_ok || sparse_ok) && lv->pairs <= 1 && lv->C >= 1 && lv->C <= 64 * kAdjMaxCJ && lv->N >= 1;
}

// fold mode: the dense layout (the target map holds features, gradients are formed from it), any C the pixel kernels take
bool adj_fold_supported(const banet_level_t* lv) {
  // (the tile kernels address a window's maps with 32-bit byte offsets built by 24-bit multiplies: pixel index, bytes per texel row)
  return adj_supported(lv) && lv->dense == 1 && lv->tgt_has_grad == 0 && (size_t)lv->H * lv->W * lv->C < ((size_t)1 << 30) &&
         lv->N < (1 << 24) && (size_t)lv->W * lv->C < ((size_t)1 << 22);
}

void adj_plan(const banet_level_t* lv, int flags, AdjPlan* pl) {
  const size_t B = lv->B, N = lv->N, C = lv->C, K = lv->K, P = 6 + K, HW = (size_t)lv->H * lv->W;   // (dense: HW == N)
  const int per = (int)((2048 + B - 1) / B);                         // ~8 waves per CU over the whole chip
  pl->fold = (flags & BANET_ADJOINT_FOLD_TARGET) ? 1 : 0;
  pl->G = (int)std::max<size_t>(1, std::min<size_t>((N + 63) / 64, (size_t)(per + kNumWaves - 1) / kNumWaves));
  pl->Ga = (int)std::max<size_t>(1, std::min<size_t>((N + 63) / 64, (size_t)((512 + B - 1) / B)));
  pl->Gm = (int)std::max<size_t>(1, std::min<size_t>((HW + 3) / 4, (size_t)((4096 + B - 1) / B)));
  pl->bigq_cap = (int)(B * N / (kSortSerial + 1) + 1);               // every cell that could hold more than kSortSerial pixels
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  pl->off_S = take(B * P * P * 4);
  pl->off_z2 = take(B * N * K * 4);
  pl->off_arec = take(B * N * 8 * 4);
  pl->off_arow = pl->fold ? 0 : take(B * N * 3 * C * 4);
  pl->off_frac = take(B * N * kFrac * 4);
  pl->off_cnt = take(B * HW * 4 + (pl->fold ? (size_t)(1 + pl->bigq_cap) * 4 : 0));   // fold: the big-cell queue follows, its counter zeroed with the counts
  pl->off_start = take(B * HW * 8);
  pl->off_cursor = take(B * HW * 4);
  pl->off_list = take(B * N * 4);
  pl->off_part = take(B * (size_t)pl->G * kNumWaves * (kAdjHdr + K) * 4);
  pl->off_chunks = take(B * ((HW + 1023) / 1024) * 4);
  pl->off_lrec = pl->fold ? take(B * N * kFrac * 4) : 0;
  pl->off_list2 = pl->fold ? take(B * N * 4) : 0;
  pl->off_lidx = pl->fold ? take(B * N * 8) : 0;
  pl->bytes = o;
}

// ---- launch of the tile kernel: (channel chunks, vector width) by C; tile shape by flags bits 4-6 (A/B) ----
template <int CJ, int V, int TW, int TH>
void launch_adj_tile_t(const AdjArgs& a, hipStream_t s) {
  const int tiles_x = (a.lv.W + TW - 1) / TW, ntiles = tiles_x * ((a.lv.H + TH - 1) / TH);
  const int total = ntiles * a.lv.B, chunk = (total + 7) / 8;
  const size_t shm = (size_t)(TW * TH + 1) * CJ * V * 64 * sizeof(float);     // + the dummy slot
  if (shm > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_tile_kernel<CJ, V, TW, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL((adj_tile_kernel<CJ, V, TW, TH>), dim3(8 * chunk), dim3(64), shm, s, a, tiles_x, ntiles, total, chunk);
}
template <int CJ, int V>
void launch_adj_tile_cv(const AdjArgs& a, int shape, hipStream_t s) {
  switch (shape) {      // LDS per wave at C = 128 / waves per CU / visits per pixel
    case 1: launch_adj_tile_t<CJ, V, 8, 4>(a, s); break;     // 16.5 KB / 9 / 2.4
    case 2: launch_adj_tile_t<CJ, V, 4, 4>(a, s); break;     //  8.5 KB / 18 / 3.1
    case 3: launch_adj_tile_t<CJ, V, 8, 2>(a, s); break;     //  8.5 KB / 18 / 3.4
    case 4: launch_adj_tile_t<CJ, V, 8, 7>(a, s); break;     // 28.5 KB / 5 / 2.0
    case 5: launch_adj_tile_t<CJ, V, 4, 2>(a, s); break;     //  4.5 KB / 32 / 4.4
    default: launch_adj_tile_t<CJ, V, 8, 4>(a, s); break;
  }
}
template <int TW, int TH>
void launch_adj_tile2_t(const AdjArgs& a, hipStream_t s) {
  const int tiles_x = (a.lv.W + TW - 1) / TW, ntiles = tiles_x * ((a.lv.H + TH - 1) / TH);
  const int total = ntiles * a.lv.B, chunk = (total + 7) / 8;
  const size_t shm = (size_t)(TW + 1) * (TH + 1) * 32 * 16;       // 512 bytes per slot, one dummy row and column
  hipLaunchKernelGGL((adj_tile2_kernel<TW, TH>), dim3(8 * chunk), dim3(64), shm, s, a, tiles_x, ntiles, total, chunk);
}

void launch_adj_tile(const AdjArgs& a, int shape, hipStream_t s) {
  const int C = a.lv.C;
  if ((C & 3) == 0 && C <= 128 && (shape == 0 || shape >= 8) && a.lv.W < 65536 && a.lv.H < 65536) {     // two visits per wave instruction
    switch (shape) {     // LDS per wave / waves per CU by LDS / visits per pixel
      case 9: launch_adj_tile2_t<8, 3>(a, s); break;      // 18.4 KB / 8 / 2.75
      case 10: launch_adj_tile2_t<8, 5>(a, s); break;     // 27.6 KB / 5 / 2.2
      case 11: launch_adj_tile2_t<8, 7>(a, s); break;     // 36.9 KB / 4 / 1.96
      case 12: launch_adj_tile2_t<16, 3>(a, s); break;    // 34.8 KB / 4 / 2.4
      case 13: launch_adj_tile2_t<16, 2>(a, s); break;    // 26.1 KB / 6 / 2.97
      default: launch_adj_tile2_t<8, 4>(a, s); break;     // 23.0 KB / 7 / 2.4
    }
    return;
  }
  if (shape >= 8) shape = 0;
  if ((C & 1) == 0) {
    if (C <= 128)
      launch_adj_tile_cv<1, 2>(a, shape, s);
    else
      launch_adj_tile_cv<2, 2>(a, shape, s);
  } else if (C <= 64) {
    launch_adj_tile_cv<1, 1>(a, shape, s);
  } else if (C <= 128) {
    launch_adj_tile_cv<2, 1>(a, shape, s);
  } else {
    launch_adj_tile_cv<4, 1>(a, shape, s);
  }
}

template <int NK>
void launch_adj_basis(const AdjArgs& a, int Ga, hipStream_t s) {
  constexpr int KP = 16 * NK, LS = KP + 20;
  const size_t shm = (size_t)KP * LS * sizeof(float);
  // > 64 KB of dynamic LDS needs the attribute; it belongs to the CURRENT device's function object, so it is set on every
  // launch (like launch_spd_solve / launch_syrk_nb) rather than once per process behind a flag
  if (shm > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_basis_kernel<NK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL((adj_basis_kernel<NK>), dim3(Ga, a.lv.B), dim3(kBlock), shm, s, a);
}

template <int NK>
void launch_adj_basis6(const AdjArgs& a, int Ga, hipStream_t s) {
  constexpr int KS = (NK + 1) / 2, NB = NK + 1;
  const size_t shm = (size_t)KS * NB * 3 * 64 * 16;
  if (a.reuse_z) {
    if (shm > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_basis6_kernel<NK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((adj_basis6_kernel<NK, true>), dim3(std::max(1, Ga / 2), a.lv.B), dim3(kAdjB6Threads), shm, s, a);
    return;
  }
  if (shm > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_basis6_kernel<NK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL((adj_basis6_kernel<NK, false>), dim3(std::max(1, Ga / 2), a.lv.B), dim3(kAdjB6Threads), shm, s, a);
}

template <int NK>
void launch_adj_basis_wide(const AdjArgs& a, int Ga, hipStream_t s) {
  constexpr int NBC = 6, KP = 16 * NK, LSC = 16 * NBC + 20;
  const size_t shm = (size_t)KP * LSC * sizeof(float);
  static_assert((size_t)KP * LSC * sizeof(float) <= 160 * 1024, "seed chunk exceeds the LDS");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&adj_basis_wide_kernel<NK, NBC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL((adj_basis_wide_kernel<NK, NBC>), dim3(Ga, a.lv.B), dim3(kBlock), shm, s, a);
}

}  // namespace

size_t dense_adjoint_workspace_bytes(const banet_level_t* lv, int flags) {
  if (!adj_supported(lv)) return 0;
  if ((flags & BANET_ADJOINT_FOLD_TARGET) && !adj_fold_supported(lv)) return 0;
  AdjPlan pl;
  adj_plan(lv, flags, &pl);
  return pl.bytes;
}

int launch_dense_adjoint(const banet_level_t* lv, const float* R, const float* T, const float* Wc, const float* gAtA,
                         const float* gAtb, const float* gabs, float* dsrc, float* dmap3, float* ddepth, float* dbasis,
                         float* dpose, int flags, void* ws, hipStream_t s) {
  if (!adj_supported(lv)) return BANET_ERR_UNSUPPORTED;
  if ((flags & BANET_ADJOINT_FOLD_TARGET) && !adj_fold_supported(lv)) return BANET_ERR_UNSUPPORTED;
  AdjPlan pl;
  adj_plan(lv, flags, &pl);
  char* base = static_cast<char*>(ws);
  const int B = lv->B, N = lv->N, K = lv->K, P = 6 + K, HW = lv->H * lv->W;
  AdjArgs a;
  a.lv = *lv;
  a.R = R;
  a.T = T;
  a.Wc = Wc;
  float* S = reinterpret_cast<float*>(base + pl.off_S);
  a.S = S;
  a.gb = gAtb;
  a.gabs = gabs;
  a.z2 = reinterpret_cast<float*>(base + pl.off_z2);
  a.arec = reinterpret_cast<float*>(base + pl.off_arec);
  a.arow = pl.fold ? nullptr : reinterpret_cast<float*>(base + pl.off_arow);
  a.frac = reinterpret_cast<float*>(base + pl.off_frac);
  a.cnt = reinterpret_cast<int*>(base + pl.off_cnt);
  a.start = reinterpret_cast<int*>(base + pl.off_start);
  a.cursor = reinterpret_cast<int*>(base + pl.off_cursor);
  a.list = reinterpret_cast<int*>(base + pl.off_list);
  a.part = reinterpret_cast<float*>(base + pl.off_part);
  a.lrec = pl.fold ? reinterpret_cast<float*>(base + pl.off_lrec) : nullptr;
  a.list2 = pl.fold ? reinterpret_cast<int*>(base + pl.off_list2) : nullptr;
  a.lidx = pl.fold ? reinterpret_cast<int2*>(base + pl.off_lidx) : nullptr;
  a.bigq = pl.fold ? a.cnt + (size_t)B * HW : nullptr;
  a.G = pl.G;
  a.dsrc = dsrc;
  a.ddepth = ddepth;
  a.dbasis = dbasis;
  a.dmap3 = dmap3;
  a.dpose = dpose;
  a.overwrite = (flags & BANET_ADJOINT_OVERWRITE) ? 1 : 0;
  a.overwrite_map = (flags & BANET_ADJOINT_OVERWRITE_MAP) ? 1 : 0;

  const size_t tot = (size_t)B * P * P;
  hipLaunchKernelGGL(adj_sym_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, gAtA, S, P, tot);
  if (hipMemsetAsync(a.cnt, 0, ((size_t)B * HW + (pl.fold ? 1 : 0)) * sizeof(int), s) != hipSuccess) return BANET_ERR_LAUNCH;
  const bool b6 = !(lv->flags & (1 << 26));   // the bf16x6 form of the GEMM-shaped piece (bit 26: fp32 MFMA, A/B)
  a.reuse_z = ((flags & BANET_ADJOINT_REUSE_DEPTH_SEED) && b6 && K > 16 && K <= 128) ? 1 : 0;     // (the other forms recompute: same result)
  switch ((K + 15) / 16) {
    case 1: launch_adj_basis<1>(a, pl.Ga, s); break;
    case 2: b6 ? launch_adj_basis6<2>(a, pl.Ga, s) : launch_adj_basis<2>(a, pl.Ga, s); break;
    case 3: b6 ? launch_adj_basis6<3>(a, pl.Ga, s) : launch_adj_basis<3>(a, pl.Ga, s); break;
    case 4: b6 ? launch_adj_basis6<4>(a, pl.Ga, s) : launch_adj_basis<4>(a, pl.Ga, s); break;
    case 5: b6 ? launch_adj_basis6<5>(a, pl.Ga, s) : launch_adj_basis<5>(a, pl.Ga, s); break;
    case 6: b6 ? launch_adj_basis6<6>(a, pl.Ga, s) : launch_adj_basis<6>(a, pl.Ga, s); break;
    case 7: b6 ? launch_adj_basis6<7>(a, pl.Ga, s) : launch_adj_basis<7>(a, pl.Ga, s); break;
    case 8: b6 ? launch_adj_basis6<8>(a, pl.Ga, s) : launch_adj_basis<8>(a, pl.Ga, s); break;
    case 9: case 10: case 11: case 12: launch_adj_basis_wide<12>(a, pl.Ga, s); break;
    case 13: case 14: case 15: case 16: launch_adj_basis_wide<16>(a, pl.Ga, s); break;
    case 0:   // pose only: no depth block -> q = zeta = e = 0; the pixel kernels read (never use) one basis / z2 element
      if (hipMemsetAsync(a.arec, 0, (size_t)B * N * 8 * sizeof(float), s) != hipSuccess) return BANET_ERR_LAUNCH;
      a.lv.basis = a.arec;
      a.z2 = a.arec;
      a.dbasis = a.arec;       // (adj_pixel2_kernel reads its rows unconditionally at clamped offsets: a valid address, never written)
      break;
    default: return BANET_ERR_UNSUPPORTED;
  }
  if (!lv->dense) {   // sparse points, [f|gx|gy] target map: one point per wave
    const int CJ = (lv->C + 63) / 64, KJ = std::max(1, (K + 63) / 64);
    const dim3 grid(pl.G, B), block(kBlock);
#define BANET_ADJ_POINT(cj, kj) \
  if (CJ == cj && KJ == kj) hipLaunchKernelGGL((adj_pixel_kernel<cj, kj, true>), grid, block, 0, s, a)
    BANET_ADJ_POINT(1, 1);
    BANET_ADJ_POINT(2, 1);
    BANET_ADJ_POINT(3, 1);
    BANET_ADJ_POINT(4, 1);
    BANET_ADJ_POINT(1, 2);
    BANET_ADJ_POINT(2, 2);
    BANET_ADJ_POINT(3, 2);
    BANET_ADJ_POINT(4, 2);
    BANET_ADJ_POINT(1, 3);
    BANET_ADJ_POINT(2, 3);
    BANET_ADJ_POINT(1, 4);
    BANET_ADJ_POINT(2, 4);
#undef BANET_ADJ_POINT
  } else if ((lv->C & 3) == 0 && (K & 3) == 0 && lv->C <= 128 && K <= 128 && !(lv->flags & (1 << 27)) &&   // bit 27: one pixel per wave (A/B)
             (size_t)N * 3 * lv->C < ((size_t)1 << 30) && N < (1 << 24)) {   // (32-bit byte offsets inside a window: its largest array is the N x 3C adjoint rows; 24-bit multiplies)
    if (a.overwrite)
      hipLaunchKernelGGL((adj_pixel2_kernel<1, true>), dim3(pl.G, B), dim3(kBlock), 0, s, a);   // (C = 256: 264 B of spills -> the one-pixel kernel)
    else
      hipLaunchKernelGGL((adj_pixel2_kernel<1, false>), dim3(pl.G, B), dim3(kBlock), 0, s, a);
  } else {
    const int CJ = (lv->C + 63) / 64, KJ = std::max(1, (K + 63) / 64);
    const dim3 grid(pl.G, B), block(kBlock);
#define BANET_ADJ_PIXEL(cj, kj) \
  if (CJ == cj && KJ == kj) hipLaunchKernelGGL((adj_pixel_kernel<cj, kj>), grid, block, 0, s, a)
    BANET_ADJ_PIXEL(1, 1);
    BANET_ADJ_PIXEL(2, 1);
    BANET_ADJ_PIXEL(3, 1);
    BANET_ADJ_PIXEL(4, 1);
    BANET_ADJ_PIXEL(1, 2);
    BANET_ADJ_PIXEL(2, 2);
    BANET_ADJ_PIXEL(3, 2);
    BANET_ADJ_PIXEL(4, 2);
    BANET_ADJ_PIXEL(1, 3);
    BANET_ADJ_PIXEL(2, 3);
    BANET_ADJ_PIXEL(3, 3);
    BANET_ADJ_PIXEL(4, 3);
    BANET_ADJ_PIXEL(1, 4);
    BANET_ADJ_PIXEL(2, 4);
    BANET_ADJ_PIXEL(3, 4);
    BANET_ADJ_PIXEL(4, 4);
#undef BANET_ADJ_PIXEL
  }
  launch_cell_scan(a.cnt, a.start, a.cursor, reinterpret_cast<int*>(base + pl.off_chunks), B, HW, s);
  hipLaunchKernelGGL(adj_fill_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, a.frac, a.cursor, a.list, N, HW);
  if (pl.fold) {
    const int2* cs = reinterpret_cast<const int2*>(a.start);
    hipLaunchKernelGGL(adj_cellsort_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, s, cs, a.list, a.frac, a.lrec, a.lidx, a.bigq, N, HW,
                       lv->W, pl.bigq_cap);
    hipLaunchKernelGGL(adj_bigcell_kernel, dim3(64), dim3(256), 0, s, cs, a.list, a.list2, a.frac, a.lrec, a.lidx, a.bigq, N, HW, lv->W,
                       pl.bigq_cap);
    launch_adj_tile(a, (flags >> 4) & 15, s);      // BANET_ADJOINT_TILE_SHAPE(k): development switch, 0 = default
  } else {
    const dim3 grid(pl.Gm, B), block(kBlock);
    launch_adj_map(a, lv->C, grid, block, s);
  }
  hipLaunchKernelGGL(adj_fold_kernel, dim3((12 + K + 127) / 128, B), dim3(128), 0, s, a.part, pl.G * kNumWaves, K, dpose);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

namespace {
struct DetPlan {
  size_t off_arow, off_frac, off_cnt, off_start, off_cursor, off_list, off_chunks, bytes;
};
void det_plan(int B, int N, int C, int H, int W, DetPlan* pl) {
  const size_t HW = (size_t)H * W;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  pl->off_arow = take((size_t)B * N * 3 * C * 4);
  pl->off_frac = take((size_t)B * N * kFrac * 4);
  pl->off_cnt = take((size_t)B * HW * 4);
  pl->off_start = take((size_t)B * HW * 8);
  pl->off_cursor = take((size_t)B * HW * 4);
  pl->off_list = take((size_t)B * N * 4);
  pl->off_chunks = take((size_t)B * ((HW + 1023) / 1024) * 4);
  pl->bytes = o;
}
}  // namespace

size_t sample_stats_grad_det_workspace_bytes(int B, int N, int C, int H, int W) {
  if (B <= 0 || N <= 0 || C < 1 || C > 64 * kAdjMaxCJ || H <= 0 || W <= 0) return 0;
  DetPlan pl;
  det_plan(B, N, C, H, W, &pl);
  return pl.bytes;
}

int launch_sample_stats_grad_det(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C,
                                 int H, int W, const float* dstats, const float* dabs, float* dconv1, float* dconv2, float* dpos,
                                 void* ws, hipStream_t s) {
  if (C < 1 || C > 64 * kAdjMaxCJ) return BANET_ERR_UNSUPPORTED;
  DetPlan pl;
  det_plan(B, N, C, H, W, &pl);
  char* base = static_cast<char*>(ws);
  const int HW = H * W;
  AdjArgs a = {};
  a.lv.B = B;
  a.lv.N = N;
  a.lv.C = C;
  a.lv.H = H;
  a.lv.W = W;
  a.arow = reinterpret_cast<float*>(base + pl.off_arow);
  a.frac = reinterpret_cast<float*>(base + pl.off_frac);
  a.cnt = reinterpret_cast<int*>(base + pl.off_cnt);
  a.start = reinterpret_cast<int*>(base + pl.off_start);
  a.cursor = reinterpret_cast<int*>(base + pl.off_cursor);
  a.list = reinterpret_cast<int*>(base + pl.off_list);
  a.dmap3 = dconv2;
  if (hipMemsetAsync(a.cnt, 0, (size_t)B * HW * sizeof(int), s) != hipSuccess) return BANET_ERR_LAUNCH;
  const int G = (N + 16 * kNumWaves - 1) / (16 * kNumWaves);
  hipLaunchKernelGGL(sstats_rows_kernel, dim3(G, B), dim3(kBlock), 0, s, conv1, conv2, px, py, N, C, H, W, dstats, dabs, dconv1,
                     dpos, a.arow, a.frac, a.cnt);
  launch_cell_scan(a.cnt, a.start, a.cursor, reinterpret_cast<int*>(base + pl.off_chunks), B, HW, s);
  hipLaunchKernelGGL(adj_fill_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, a.frac, a.cursor, a.list, N, HW);
  {
    const int Gm = (int)std::max<size_t>(1, std::min<size_t>(((size_t)HW + 3) / 4, (size_t)((4096 + B - 1) / B)));
    const dim3 grid(Gm, B), block(kBlock);
    launch_adj_map(a, C, grid, block, s);
  }
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

int launch_target_map_adjoint(const float* dmap3, float* dimg, int B, int H, int W, int C, int overwrite, hipStream_t s) {
  const size_t total = (size_t)B * H * W * C;
  if (total == 0) return BANET_OK;
  if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(dmap3) | reinterpret_cast<uintptr_t>(dimg)) & 15) == 0) {
    const size_t total4 = total / 4;
    hipLaunchKernelGGL(target_map_adjoint4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, dmap3, dimg, H, W, C / 4,
                       total4, overwrite);
    return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(target_map_adjoint_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dmap3, dimg, H, W, C, total, overwrite);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
