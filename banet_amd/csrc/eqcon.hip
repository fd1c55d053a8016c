// EquationConstruction / EquationConstructionGrad for gfx950 -- the API-parity kernels for
// the reference's TF custom ops (utils.cu:150-171,219-417 and :420-428,465-694).
//
// Forward.  AtA = sum_n J_n^T (G_n^T G_n) J_n,  Atb = sum_n J_n^T G_n^T d_n, computed per
// 32-pixel tile without materialising any per-pixel PxP matrix (utils.cu:356-365 does):
//   A  J tile (64 rows x P) -> LDS, coalesced
//   B  M_n = G_n^T G_n (2x2), g_n = G_n^T d_n: one pixel per wave at a time, lane = channel
//      pair (coalesced rows of G and d), transposing-butterfly reduction
//   C  AtA += Z^T J  with Z_n = M_n J_n formed on the fly: a 64-deep SYRK-like update on
//      v_mfma_f32_16x16x4_f32, upper-triangular 16x16 blocks only, accumulators in registers
//   D  Atb += J_n^T g_n, lane = column
// Partials are reduced in fixed order by ba_reduce_kernel (assemble.hip).
#include <cstdlib>

#include "kernels.hpp"

namespace banet {

constexpr int kEqPix = 32;

// fixed-order sum of the per-workgroup partials [P*P + P] (cf. utils.cu:181-198)
__global__ __launch_bounds__(256) void eq_reduce_kernel(const float* __restrict__ partials, int G, int pstride, int P,
                                                        float* __restrict__ AtA, float* __restrict__ Atb) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P * P + P) return;
  const float* p = partials + (size_t)b * G * pstride + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = 0;
  for (; i + 3 < G; i += 4) {
    s0 += p[(size_t)(i + 0) * pstride];
    s1 += p[(size_t)(i + 1) * pstride];
    s2 += p[(size_t)(i + 2) * pstride];
    s3 += p[(size_t)(i + 3) * pstride];
  }
  for (; i < G; ++i) s0 += p[(size_t)i * pstride];
  const float v = (s0 + s1) + (s2 + s3);
  if (e < P * P)
    AtA[(size_t)b * P * P + e] = v;
  else
    Atb[(size_t)b * P + (e - P * P)] = v;
}

void launch_reduce(const float* partials, int B, int G, int pstride, int P, float* AtA, float* Atb, hipStream_t s) {
  const int total = P * P + P;
  hipLaunchKernelGGL(eq_reduce_kernel, dim3((total + 255) / 256, B), dim3(256), 0, s, partials, G, pstride, P, AtA, Atb);
}
  // pixels per tile (64 Jacobian rows = 16 MFMA k-steps)

struct EqArgs {
  const float* J;
  const float* G;
  const float* d;
  float* partials;
  int B, N, C, P, Gr, tiles, pstride;
};

template <int NB>
__global__ __launch_bounds__(kBlock) void eq_construction_kernel(const EqArgs a) {
  constexpr int PP = NB * 16 + 16;                 // LDS row stride; bank shift 16 per row
  constexpr int NBLK = NB * (NB + 1) / 2;
  constexpr int SLOTS = (NBLK + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sJ = smem;                    // [64][PP]
  float* sM = sJ + 2 * kEqPix * PP;    // [32][4]  m11,m12,m22,-
  float* sg = sM + kEqPix * 4;         // [32][2]
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id();
  const int N = a.N, C = a.C, P = a.P;
  const bool vec2 = (C & 1) == 0;

  f32x4 acc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  float atb0 = 0.f, atb1 = 0.f;  // columns tid and tid+256

  // zero the padding columns once (they are never overwritten by the tile loads)
  for (int idx = tid; idx < 2 * kEqPix * PP; idx += kBlock) sJ[idx] = 0.f;
  __syncthreads();

  for (int t = g; t < a.tiles; t += a.Gr) {
    const int pt0 = t * kEqPix;
    const int npx = min(kEqPix, N - pt0);
    // ---- A: J tile -------------------------------------------------------------------
    const float* Jt = a.J + ((size_t)b * N + pt0) * 2 * P;
    for (int idx = tid; idx < 2 * kEqPix * P; idx += kBlock) {
      const int row = idx / P, col = idx - row * P;
      sJ[row * PP + col] = (row < 2 * npx) ? Jt[idx] : 0.f;
    }
    // ---- B: channel reductions, wave w owns pixels 8w..8w+7 ---------------------------
    auto pixel_q5 = [&](int n) -> Q5 {
      Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
      if (n >= npx) return q;  // wave-uniform
      const size_t base = ((size_t)b * N + pt0 + n) * C;
      if (vec2) {
        for (int c = lane * 2; c < C; c += 128) {
          const float4 gg = *reinterpret_cast<const float4*>(a.G + (base + c) * 2);
          const float2 dd = *reinterpret_cast<const float2*>(a.d + base + c);
          q.m11 = fmaf(gg.x, gg.x, q.m11);
          q.m12 = fmaf(gg.x, gg.y, q.m12);
          q.m22 = fmaf(gg.y, gg.y, q.m22);
          q.g1 = fmaf(gg.x, dd.x, q.g1);
          q.g2 = fmaf(gg.y, dd.x, q.g2);
          q.m11 = fmaf(gg.z, gg.z, q.m11);
          q.m12 = fmaf(gg.z, gg.w, q.m12);
          q.m22 = fmaf(gg.w, gg.w, q.m22);
          q.g1 = fmaf(gg.z, dd.y, q.g1);
          q.g2 = fmaf(gg.w, dd.y, q.g2);
        }
      } else {
        for (int c = lane; c < C; c += 64) {
          const float2 gg = *reinterpret_cast<const float2*>(a.G + (base + c) * 2);
          const float dd = a.d[base + c];
          q.m11 = fmaf(gg.x, gg.x, q.m11);
          q.m12 = fmaf(gg.x, gg.y, q.m12);
          q.m22 = fmaf(gg.y, gg.y, q.m22);
          q.g1 = fmaf(gg.x, dd, q.g1);
          q.g2 = fmaf(gg.y, dd, q.g2);
        }
      }
      return q;
    };
    {
      const int base = 8 * w;
      auto L1 = [&](int o) { return q5_merge(pixel_q5(base + o), pixel_q5(base + o + 1), 32); };
      auto L2 = [&](int o) { return q5_merge(L1(o), L1(o + 2), 16); };
      Q5 q = q5_merge(L2(0), L2(4), 8);
      q.m11 += dpp_mov<kDppHalfMirror>(q.m11);  // remaining lane bits 2,1,0 (i^7, i^2, i^1 cover all 8)
      q.m12 += dpp_mov<kDppHalfMirror>(q.m12);
      q.m22 += dpp_mov<kDppHalfMirror>(q.m22);
      q.g1 += dpp_mov<kDppHalfMirror>(q.g1);
      q.g2 += dpp_mov<kDppHalfMirror>(q.g2);
      q = q5_finish(q);
      if ((lane & 7) == 0) {
        const int n = base + (((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2));
        sM[n * 4 + 0] = q.m11;
        sM[n * 4 + 1] = q.m12;
        sM[n * 4 + 2] = q.m22;
        sg[n * 2 + 0] = q.g1;
        sg[n * 2 + 1] = q.g2;
      }
    }
    __syncthreads();
    // ---- C: AtA += Z^T J on the matrix cores ------------------------------------------
    {
      const int col = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int tb = s * 4 + w;  // wave-uniform flat upper-block index
        if (tb < NBLK) {
          int bi = 0, rem = tb;
          while (rem >= NB - bi) {
            rem -= NB - bi;
            ++bi;
          }
          const int bj = bi + rem;
          f32x4 c4 = acc[s];
#pragma unroll 4
          for (int kk = 0; kk < 2 * kEqPix / 4; ++kk) {
            const int row = 4 * kk + kq, n = row >> 1, r = row & 1;
            const float m0 = sM[n * 4 + r], m1 = sM[n * 4 + r + 1];  // (m11,m12) or (m12,m22)
            const float z = m0 * sJ[(2 * n) * PP + 16 * bi + col] + m1 * sJ[(2 * n + 1) * PP + 16 * bi + col];
            const float bv = sJ[row * PP + 16 * bj + col];
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(z, bv, c4, 0, 0, 0);
          }
          acc[s] = c4;
        }
      }
    }
    // ---- D: Atb += J^T g ---------------------------------------------------------------
    for (int n = 0; n < npx; ++n) {
      const float g1 = sg[n * 2], g2 = sg[n * 2 + 1];
      if (tid < P) atb0 += sJ[(2 * n) * PP + tid] * g1 + sJ[(2 * n + 1) * PP + tid] * g2;
      if (tid + kBlock < P) atb1 += sJ[(2 * n) * PP + tid + kBlock] * g1 + sJ[(2 * n + 1) * PP + tid + kBlock] * g2;
    }
    __syncthreads();
  }
  // ---- epilogue -----------------------------------------------------------------------
  float* __restrict__ part = a.partials + ((size_t)b * a.Gr + g) * a.pstride;
  {
    const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int tb = s * 4 + w;
      if (tb < NBLK) {
        int bi = 0, rem = tb;
        while (rem >= NB - bi) {
          rem -= NB - bi;
          ++bi;
        }
        const int bj = bi + rem;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 16 * bi + rq + r, cc = 16 * bj + col;
          if (rr < P && cc < P && (bj > bi || rr <= cc)) {
            part[rr * P + cc] = acc[s][r];
            part[cc * P + rr] = acc[s][r];
          }
        }
      }
    }
  }
  if (tid < P) part[P * P + tid] = atb0;
  if (tid + kBlock < P) part[P * P + tid + kBlock] = atb1;
}

// --------------------------------------------------------------------------------------
// Backward:  dA_n = 2 (G_n J_n) g0 + d_n g1^T, never forming the CxP matrices:
//   U = J g0 (2xP), v = J g1 (2), Q = U J^T (2x2), M = G^T G, g = G^T d
//   dJ = 2 M U + g g1^T ;  dG = 2 G Q + d v^T ;  dd = G v
// --------------------------------------------------------------------------------------
struct EqGradArgs {
  const float* J;
  const float* G;
  const float* d;
  const float* g0;
  const float* g1;
  float* gJ;
  float* gG;
  float* gd;
  int B, N, C, P, tiles;
};

__global__ __launch_bounds__(kBlock) void eq_construction_grad_kernel(const EqGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int P = a.P, C = a.C, N = a.N;
  const int PP = P + 1;
  float* sJ = smem;                    // [64][PP]
  float* sU = sJ + 2 * kEqPix * PP;    // [64][PP]
  float* sg1 = sU + 2 * kEqPix * PP;   // [P]
  float* sQ = sg1 + ((P + 3) & ~3);    // [32][12]: m11,m12,m22,g1,g2,-, q00,q01,q10,q11, v0,v1
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id();
  const int pt0 = t * kEqPix;
  const int npx = min(kEqPix, N - pt0);
  const float* g0 = a.g0 + (size_t)b * P * P;
  for (int j = tid; j < P; j += kBlock) sg1[j] = a.g1[(size_t)b * P + j];
  const float* Jt = a.J + ((size_t)b * N + pt0) * 2 * P;
  for (int idx = tid; idx < 2 * kEqPix * P; idx += kBlock) {
    const int row = idx / P, col = idx - row * P;
    sJ[row * PP + col] = (row < 2 * npx) ? Jt[idx] : 0.f;
  }
  __syncthreads();
  // U = J g0 : thread owns column j, 8 rows at a time in registers
  for (int j = tid; j < P; j += kBlock) {
    for (int r0 = 0; r0 < 2 * kEqPix; r0 += 8) {
      float u[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = 0.f;
      for (int i = 0; i < P; ++i) {
        const float gv = g0[(size_t)i * P + j];
#pragma unroll
        for (int r = 0; r < 8; ++r) u[r] = fmaf(sJ[(r0 + r) * PP + i], gv, u[r]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) sU[(r0 + r) * PP + j] = u[r];
    }
  }
  __syncthreads();
  // per pixel: M, g (channels), Q, v (columns); wave w owns pixels 8w..8w+7
  for (int i = 0; i < 8; ++i) {
    const int n = 8 * w + i;
    if (n >= npx) break;
    const size_t base = ((size_t)b * N + pt0 + n) * C;
    float m11 = 0.f, m12 = 0.f, m22 = 0.f, q1 = 0.f, q2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float gx = a.G[(base + c) * 2], gy = a.G[(base + c) * 2 + 1], dd = a.d[base + c];
      m11 = fmaf(gx, gx, m11);
      m12 = fmaf(gx, gy, m12);
      m22 = fmaf(gy, gy, m22);
      q1 = fmaf(gx, dd, q1);
      q2 = fmaf(gy, dd, q2);
    }
    float q00 = 0.f, q01 = 0.f, q10 = 0.f, q11 = 0.f, v0 = 0.f, v1 = 0.f;
    for (int j = lane; j < P; j += 64) {
      const float j0 = sJ[(2 * n) * PP + j], j1 = sJ[(2 * n + 1) * PP + j];
      const float u0 = sU[(2 * n) * PP + j], u1 = sU[(2 * n + 1) * PP + j];
      q00 = fmaf(u0, j0, q00);
      q01 = fmaf(u0, j1, q01);
      q10 = fmaf(u1, j0, q10);
      q11 = fmaf(u1, j1, q11);
      v0 = fmaf(sg1[j], j0, v0);
      v1 = fmaf(sg1[j], j1, v1);
    }
    m11 = wave_sum(m11);
    m12 = wave_sum(m12);
    m22 = wave_sum(m22);
    q1 = wave_sum(q1);
    q2 = wave_sum(q2);
    q00 = wave_sum(q00);
    q01 = wave_sum(q01);
    q10 = wave_sum(q10);
    q11 = wave_sum(q11);
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    if (lane == 0) {
      float* o = sQ + n * 12;
      o[0] = m11; o[1] = m12; o[2] = m22; o[3] = q1; o[4] = q2;
      o[6] = q00; o[7] = q01; o[8] = q10; o[9] = q11; o[10] = v0; o[11] = v1;
    }
    // dG, dd for this pixel (coalesced over channels)
    for (int c = lane; c < C; c += 64) {
      const float gx = a.G[(base + c) * 2], gy = a.G[(base + c) * 2 + 1], dd = a.d[base + c];
      a.gG[(base + c) * 2] = 2.f * (gx * q00 + gy * q10) + dd * v0;
      a.gG[(base + c) * 2 + 1] = 2.f * (gx * q01 + gy * q11) + dd * v1;
      a.gd[base + c] = gx * v0 + gy * v1;
    }
  }
  __syncthreads();
  // dJ = 2 M U + g g1^T
  float* gJt = a.gJ + ((size_t)b * N + pt0) * 2 * P;
  for (int idx = tid; idx < 2 * npx * P; idx += kBlock) {
    const int row = idx / P, j = idx - row * P, n = row >> 1, r = row & 1;
    const float* o = sQ + n * 12;
    const float mr0 = o[r], mr1 = o[r + 1];
    gJt[idx] = 2.f * (mr0 * sU[(2 * n) * PP + j] + mr1 * sU[(2 * n + 1) * PP + j]) + o[3 + r] * sg1[j];
  }
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
static int eq_nb(int P) {
  const int nb = (P + 15) / 16;
  if (nb <= 1) return 1;
  if (nb <= 2) return 2;
  if (nb <= 3) return 3;
  if (nb <= 5) return 5;
  if (nb <= 9) return 9;
  if (nb <= 17) return 17;
  if (nb <= 19) return 19;   // 272 < P <= 304 (cfg-5's 8-frame windows: P = 298): eqcon_syrk.hip's 17-block pass + three jobs for blocks 17, 18
  return -1;
}

int plan_eq(int B, int N, int C, int P, EqPlan* pl) {
  if (B <= 0 || N <= 0 || C <= 0 || P <= 0) return BANET_ERR_INVALID_ARG;
  pl->nb = eq_nb(P);
  if (pl->nb < 0) return BANET_ERR_UNSUPPORTED;
  pl->tiles = (N + kEqPix - 1) / kEqPix;
  int target = (512 + B - 1) / B;
  int G = pl->tiles / 2;
  if (G > target) G = target;
  if (G < 1) G = 1;
  // P <= 144: the SYRK formulation (eqcon_syrk.hip), one workgroup per CU, >= 1 step of 16 pixels per wave
  // (environment BANET_EQ_LDS_KERNEL=1 keeps the LDS-operand kernel: development A/B only)
  static const bool force_lds = getenv("BANET_EQ_LDS_KERNEL") != nullptr && atoi(getenv("BANET_EQ_LDS_KERNEL")) != 0;
  pl->fast = (pl->nb <= 19 && !force_lds) ? 1 : 0;
  if (pl->fast) {
    G = (256 + B - 1) / B;
    const int steps = (N + 15) / 16;
    if (G > (steps + 3) / 4) G = (steps + 3) / 4;
    if (G < 1) G = 1;
  }
  pl->Gr = G;
  pl->pstride = (int)align_up((size_t)P * P + P, 4);
  pl->partial_bytes = align_up((size_t)B * G * pl->pstride * sizeof(float), 256);
  pl->off_rec = pl->partial_bytes;
  if (pl->fast) pl->partial_bytes += eq_syrk_record_bytes(B, N);
  return BANET_OK;
}

template <int NB>
static void launch_eq_nb(const EqArgs& a, hipStream_t s) {
  const size_t lds = ((size_t)2 * kEqPix * (NB * 16 + 16) + kEqPix * 6) * sizeof(float);
  auto k = eq_construction_kernel<NB>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(a.Gr, a.B), dim3(kBlock), lds, s, a);
}

int launch_eq(const float* J, const float* G, const float* d, float* AtA, float* Atb, int B, int N, int C, int P,
              const EqPlan& pl, float* partials, hipStream_t s) {
  if (pl.fast) {
    const int rc = launch_eq_syrk(J, G, d, B, N, C, P, pl.nb, pl.Gr, pl.pstride, partials,
                                  reinterpret_cast<float*>(reinterpret_cast<char*>(partials) + pl.off_rec), s);
    if (rc != BANET_OK) return rc;
    launch_reduce(partials, B, pl.Gr, pl.pstride, P, AtA, Atb, s);
    return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
  }
  EqArgs a{J, G, d, partials, B, N, C, P, pl.Gr, pl.tiles, pl.pstride};
  switch (pl.nb) {
    case 1: launch_eq_nb<1>(a, s); break;
    case 2: launch_eq_nb<2>(a, s); break;
    case 3: launch_eq_nb<3>(a, s); break;
    case 5: launch_eq_nb<5>(a, s); break;
    case 9: launch_eq_nb<9>(a, s); break;
    case 17: launch_eq_nb<17>(a, s); break;
    case 19: launch_eq_nb<19>(a, s); break;
    default: return BANET_ERR_UNSUPPORTED;
  }
  launch_reduce(partials, B, pl.Gr, pl.pstride, P, AtA, Atb, s);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

int launch_eq_grad(const float* J, const float* G, const float* d, const float* g0, const float* g1, float* gJ,
                   float* gG, float* gd, int B, int N, int C, int P, hipStream_t s) {
  if (B <= 0 || N <= 0 || C <= 0 || P <= 0) return BANET_ERR_INVALID_ARG;
  const size_t lds = ((size_t)4 * kEqPix * (P + 1) + ((P + 3) & ~3) + kEqPix * 12) * sizeof(float);
  if (lds > 160 * 1024) return BANET_ERR_UNSUPPORTED;
  EqGradArgs a{J, G, d, g0, g1, gJ, gG, gd, B, N, C, P, (N + kEqPix - 1) / kEqPix};
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)eq_construction_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(eq_construction_grad_kernel, dim3(a.tiles, B), dim3(kBlock), lds, s, a);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
