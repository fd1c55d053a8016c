// EquationConstructionGrad for P <= 304 on the bf16 matrix pipe (the literal op of utils.cu:420-428,465-694):
//   dJ = 2 M U + g g1^T,  dG = 2 G Q + d v^T,  dd = G v    with  U = J g0 (2 x P per pixel),  Q = U J^T (2x2),  v = J g1 (2),
//   M = G^T G, g = G^T d.
// U is the only large product: the [2N x P] . [P x P] GEMM.  Three kernels:
//   eq_pixel_records_kernel (eqcon_syrk.hip, raw mode)  streams G and d -> (m11, m12, m22, g1, g2) per pixel;
//   eq_grad_u_kernel<NB>   one 512-thread workgroup per CU; g0 is split ONCE per workgroup into three bf16 pieces laid out in
//                          LDS as MFMA B operands (P <= 144: 138 KB); a wave takes 16 rows of J (8 pixels) at a time, splits
//                          each 16 x 32 piece of them exactly into three bf16 pieces (registers) and accumulates U = J g0 with six
//                          v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block (fp32 accumulate, products exact); the epilogue is
//                          lane-local: both rows of a pixel sit in one lane's accumulator registers, so dJ is written from
//                          there, and the pixel's Q and v are reduced over the 16 lanes of a DPP row;
//   eq_grad_gd_kernel      streams G and d again with (Q, v) per pixel -> dG, dd.
// The first-generation kernel (eq_construction_grad_kernel, eqcon.hip: J and U tiles in LDS, U on the VALU) stays for
// callers that pass no workspace.
#include "kernels.hpp"
#include "syrk_split.hpp"

namespace banet {

namespace {
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
constexpr int kGradThreads = 512, kGradWaves = kGradThreads / 64;   // 2 waves per SIMD: 256 registers each

}  // namespace

struct EqGradUArgs {
  const float* J;      // [B][N][2][P]
  const float* g0;     // [B][P][P]
  const float* g1;     // [B][P]
  const float* rec;    // [B][N][8]: m11, m12, m22, g1, g2
  float* gJ;           // [B][N][2][P]
  float* rec2;         // [B][N][8]: q00, q01, q10, q11, v0, v1
  int N, P, Gr;
  int ob0;             // first 16-column block of this pass's output columns
  int accumulate;      // 1: add this pass's share of (Q, v) to rec2 (passes after the first one)
};

// NBK = 16-row blocks of g0 (the inner dimension, all of P), NB = 16-column blocks of this pass's output columns.  P <= 144:
// one pass <9, 9>.  144 < P <= 272: g0's split image for all 17 x 17 blocks (413 KB) does not fit the LDS, so the output
// columns are cut into chunks of 5 blocks (9 k-steps x 5 x 3 KB = 135 KB): four passes over J, (Q, v) accumulated across them.
template <int NBK, int NB>
__global__ __launch_bounds__(kGradThreads) void eq_grad_u_kernel(const EqGradUArgs a) {
  constexpr int KS = (NBK + 1) / 2;                      // 32-wide k steps covering 16 NBK rows of g0
  extern __shared__ __attribute__((aligned(16))) unsigned sB[];   // [KS][NB][3][64] quads: g0 as MFMA B operands
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N, P = a.P;
  const int m = lane & 15, kq = lane >> 4;
  const float* __restrict__ J_b = a.J + (size_t)b * N * 2 * P;
  const float* __restrict__ g0 = a.g0 + (size_t)b * P * P;
  const float* __restrict__ rec_b = a.rec + (size_t)b * N * 8;
  float* __restrict__ gJ_b = a.gJ + (size_t)b * N * 2 * P;
  float* __restrict__ rec2_b = a.rec2 + (size_t)b * N * 8;

  // ---- g0 -> LDS, split into bf16 pieces in operand layout: lane (n = m, kq) holds k = 32 ks + 8 kq + e, column 16 bj + n
  for (int task = w; task < KS * NB; task += kGradWaves) {
    const int ks = task / NB, bj = task - ks * NB;
    float vv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * ks + 8 * kq + e, c = 16 * (a.ob0 + bj) + m;
      vv[e] = (k < P && c < P) ? g0[(size_t)k * P + c] : 0.f;
    }
    u32x4_t pc[3];
    split8_bf16x3(vv, pc);
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x4_t*>(&sB[(((ks * NB + bj) * 3 + t) * 64 + lane) * 4]) = pc[t];
  }
  float g1v[NB];
#pragma unroll
  for (int bj = 0; bj < NB; ++bj) g1v[bj] = (16 * (a.ob0 + bj) + m < P) ? a.g1[(size_t)b * P + 16 * (a.ob0 + bj) + m] : 0.f;
  __syncthreads();

  // ---- blocks of 16 rows (8 pixels): block index rb over the window, this wave's share
  const int nblk = (N + 7) >> 3, nwaves = a.Gr * kGradWaves, gw = g * kGradWaves + w;
  const int rb0 = (int)(((long long)nblk * gw) / nwaves), rb1 = (int)(((long long)nblk * (gw + 1)) / nwaves);
  const bool even = (P & 1) == 0;
  auto load_a = [&](int rb, int ks, float (&v)[8]) __attribute__((always_inline)) {
    // A operand: lane (i = m, kq) holds J[row 16 rb + m][32 ks + 8 kq + e]; the 4 kq lanes of a row read 128 contiguous bytes
    const int row = min(16 * rb + m, 2 * N - 1);
    const int c0 = 32 * ks + 8 * kq;
    const float* p = J_b + (size_t)row * P + c0;
    if (even && c0 + 8 <= P) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float2 t = *reinterpret_cast<const float2*>(p + 2 * e);
        v[2 * e] = t.x;
        v[2 * e + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (c0 + e < P) ? p[e] : 0.f;
    }
    if (16 * rb + m >= 2 * N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
  };

  for (int rb = rb0; rb < rb1; ++rb) {
    f32x4 acc[NB];
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) acc[bj] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av[8], an[8];
    load_a(rb, 0, av);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("" ::: "memory");   // the B operands are re-read from LDS every block: hoisting all 45 x 3 quads out of the
                                       // row-block loop (they are loop-invariant) spilled 2.3 KB per lane
      u32x4_t pa[3];
      split8_bf16x3(av, pa);
      if (ks + 1 < KS) load_a(rb, ks + 1, an);             // next k step's rows in flight during this step's MFMAs
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) {
        u32x4_t pb[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) pb[t] = *reinterpret_cast<const u32x4_t*>(&sB[(((ks * NB + bj) * 3 + t) * 64 + lane) * 4]);
        f32x4 c = acc[bj];
        constexpr int kTa[6] = {2, 0, 1, 1, 0, 0}, kTb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
        for (int t = 0; t < 6; ++t)
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8g, pa[kTa[t]]), __builtin_bit_cast(bf16x8g, pb[kTb[t]]), c, 0,
                                                      0, 0);
        acc[bj] = c;
      }
      if (ks + 1 < KS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = an[e];
      }
    }
    // ---- epilogue: lane (m, kq) holds U[row 16 rb + 4 kq + r][col 16 bj + m]: rows r = 0,1 -> pixel 8 rb + 2 kq, r = 2,3 -> the next
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      const int n = 8 * rb + 2 * kq + hq;                  // pixel
      const bool ok = n < N;
      const size_t nn = (size_t)min(n, N - 1);
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(rec_b + nn * 8);
      const float m11 = r0[0], m12 = r0[1], m22 = r0[2], gg1 = r0[3], gg2 = rec_b[nn * 8 + 4];
      float q00 = 0.f, q01 = 0.f, q10 = 0.f, q11 = 0.f, v0 = 0.f, v1 = 0.f;
      const float* jr = J_b + nn * 2 * P;
      float* gr = gJ_b + nn * 2 * P;
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) {
        const int c = 16 * (a.ob0 + bj) + m;
        if (c < P) {
          const float u0 = acc[bj][2 * hq], u1 = acc[bj][2 * hq + 1];
          const float j0 = jr[c], j1 = jr[P + c];
          if (ok) {
            gr[c] = 2.f * (m11 * u0 + m12 * u1) + gg1 * g1v[bj];
            gr[P + c] = 2.f * (m12 * u0 + m22 * u1) + gg2 * g1v[bj];
          }
          q00 = fmaf(u0, j0, q00);
          q01 = fmaf(u0, j1, q01);
          q10 = fmaf(u1, j0, q10);
          q11 = fmaf(u1, j1, q11);
          v0 = fmaf(g1v[bj], j0, v0);
          v1 = fmaf(g1v[bj], j1, v1);
        }
      }
      q00 = row16_sum(q00);
      q01 = row16_sum(q01);
      q10 = row16_sum(q10);
      q11 = row16_sum(q11);
      v0 = row16_sum(v0);
      v1 = row16_sum(v1);
      if (ok && m == 0) {
        float4* o = reinterpret_cast<float4*>(rec2_b + nn * 8);
        if (a.accumulate) {   // same lane, passes in launch order: a fixed summation order
          const float4 p0 = o[0], p1 = o[1];
          o[0] = make_float4(p0.x + q00, p0.y + q01, p0.z + q10, p0.w + q11);
          o[1] = make_float4(p1.x + v0, p1.y + v1, 0.f, 0.f);
        } else {
          o[0] = make_float4(q00, q01, q10, q11);
          o[1] = make_float4(v0, v1, 0.f, 0.f);
        }
      }
    }
  }
}

// ---- dG = 2 G Q + d v^T, dd = G v ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void eq_grad_gd_kernel(const float* __restrict__ G, const float* __restrict__ d,
                                                            const float* __restrict__ rec2, int N, int C, float* __restrict__ gG,
                                                            float* __restrict__ gd) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = wave_id();
  const bool vec2 = (C & 1) == 0;
  for (int i = 0; i < 8; ++i) {
    const int n = blockIdx.x * 32 + 8 * w + i;
    if (n >= N) break;                                    // wave-uniform
    const size_t q = (size_t)b * N + n, base = q * C;
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rec2 + q * 8);
    const float q00 = r0[0], q01 = r0[1], q10 = r0[2], q11 = r0[3], v0 = rec2[q * 8 + 4], v1 = rec2[q * 8 + 5];
    if (vec2) {
      for (int c = lane * 2; c < C; c += 128) {
        const float4 gg = *reinterpret_cast<const float4*>(G + (base + c) * 2);
        const float2 dd = *reinterpret_cast<const float2*>(d + base + c);
        float4 o;
        o.x = 2.f * (gg.x * q00 + gg.y * q10) + dd.x * v0;
        o.y = 2.f * (gg.x * q01 + gg.y * q11) + dd.x * v1;
        o.z = 2.f * (gg.z * q00 + gg.w * q10) + dd.y * v0;
        o.w = 2.f * (gg.z * q01 + gg.w * q11) + dd.y * v1;
        *reinterpret_cast<float4*>(gG + (base + c) * 2) = o;
        *reinterpret_cast<float2*>(gd + base + c) = make_float2(gg.x * v0 + gg.y * v1, gg.z * v0 + gg.w * v1);
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float gx = G[(base + c) * 2], gy = G[(base + c) * 2 + 1], dd = d[base + c];
        gG[(base + c) * 2] = 2.f * (gx * q00 + gy * q10) + dd * v0;
        gG[(base + c) * 2 + 1] = 2.f * (gx * q01 + gy * q11) + dd * v1;
        gd[base + c] = gx * v0 + gy * v1;
      }
    }
  }
}

size_t eq_grad_fast_ws_bytes(int B, int N, int P) {
  if (P > 304) return 0;
  return 2 * align_up((size_t)B * N * 8 * sizeof(float), 256);
}

void launch_eq_pixel_records(const float* G, const float* d, int B, int N, int C, int raw, float* rec, hipStream_t s);

template <int NBK, int NB = NBK>
static void launch_u(const EqGradUArgs& a, int B, hipStream_t s) {
  constexpr int KS = (NBK + 1) / 2;
  const size_t lds = (size_t)KS * NB * 3 * 64 * 16;
  auto k = eq_grad_u_kernel<NBK, NB>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(a.Gr, B), dim3(kGradThreads), lds, s, a);
}

int launch_eq_grad_fast(const float* J, const float* G, const float* d, const float* g0, const float* g1, float* gJ, float* gG,
                        float* gd, int B, int N, int C, int P, void* ws, hipStream_t s) {
  float* rec = static_cast<float*>(ws);
  float* rec2 = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)B * N * 8 * sizeof(float), 256));
  launch_eq_pixel_records(G, d, B, N, C, 1, rec, s);
  int Gr = (256 + B - 1) / B;
  const int nblk = (N + 7) / 8;
  if (Gr > (nblk + kGradWaves - 1) / kGradWaves) Gr = (nblk + kGradWaves - 1) / kGradWaves;
  if (Gr < 1) Gr = 1;
  EqGradUArgs a{J, g0, g1, rec, gJ, rec2, N, P, Gr, 0, 0};
  const int nb = (P + 15) / 16;
  if (nb <= 1)
    launch_u<1>(a, B, s);
  else if (nb <= 3)
    launch_u<3>(a, B, s);
  else if (nb <= 5)
    launch_u<5>(a, B, s);
  else if (nb <= 9)
    launch_u<9>(a, B, s);
  else if (nb <= 17) {   // output columns in chunks of 5 blocks: 0-4, 5-9, 10-14, 15-16
    for (int ob = 0; ob < nb; ob += 5) {
      a.ob0 = ob;
      a.accumulate = ob > 0;
      if (nb - ob > 2)
        launch_u<17, 5>(a, B, s);     // columns >= P are masked
      else
        launch_u<17, 2>(a, B, s);
    }
  } else if (nb <= 19) {   // 272 < P <= 304 (round 5; cfg-5's P = 298): 10 k-steps x 5 blocks x 3 KB = 150 KB of split g0 per pass, four passes
    for (int ob = 0; ob < nb; ob += 5) {      // output columns 0-4, 5-9, 10-14, 15-18 (blocks / columns >= P are masked)
      a.ob0 = ob;
      a.accumulate = ob > 0;
      launch_u<19, 5>(a, B, s);
    }
  } else
    return BANET_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(eq_grad_gd_kernel, dim3((N + 31) / 32, B), dim3(kBlock), 0, s, G, d, rec2, N, C, gG, gd);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
