// EquationConstruction forward for P <= 272 on the SYRK engine of syrk.hip (the literal op of utils.cu:150-171,219-417).
//   AtA = sum_n J_n^T M_n J_n,  Atb = sum_n J_n^T g_n,   M_n = G_n^T G_n (2x2, PSD),  g_n = G_n^T d_n
// M = L L^T (2x2 Cholesky) turns the sum into a plain W^T W over the 2N rows  w = L^T J :
//   w_{2n} = l11 j_{2n} + l21 j_{2n+1},  w_{2n+1} = l22 j_{2n+1},   AtA = sum_r w_r w_r^T,
// and since g lies in the range of M (= range of L), Atb = sum_r h_r w_r with h = L^{-1} g (zero pivots give zero
// rows and zero h).  Two kernels, like the fused path's gather / syrk split:
//   eq_pixel_records_kernel   streams G and d (3C floats per pixel, 59 % of the op's bytes) at high occupancy and
//                             leaves 8 floats per pixel: l11, l21, l22, h0, h1;
//   eq_syrk_kernel<NB>        one wave per SIMD, operands straight from registers, A and B operands are the SAME
//                             values w split exactly into three bf16 pieces: six v_mfma_f32_16x16x32_bf16 per 16x16
//                             block and 32 rows, fp32 accumulate (syrk_split.hpp); Atb rides along as one record
//                             block row (A operand row 0 = h).
// 144 < P <= 272 (17 column blocks) runs the same engine as four jobs -- two diagonal groups and two off-diagonal
// rectangles, four passes over J.  The previous kernel (eq_construction_kernel, eqcon.hip: J tile in LDS, every fp32-MFMA
// operand fetched from LDS, two barriers per 32 pixels) stays behind BANET_EQ_LDS_KERNEL=1 for A/B.
#include "kernels.hpp"
#include "syrk_split.hpp"

namespace banet {

namespace {
typedef __bf16 bf16x8e __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mm6e(const u32x4_t (&x)[3], const u32x4_t (&y)[3], f32x4 c) {
  constexpr int kTa[6] = {2, 0, 1, 1, 0, 0}, kTb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
  for (int t = 0; t < 6; ++t)
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8e, x[kTa[t]]), __builtin_bit_cast(bf16x8e, y[kTb[t]]), c, 0, 0,
                                                0);
  return c;
}
}  // namespace

// ---- per-pixel records -----------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void eq_pixel_records_kernel(const float* __restrict__ G, const float* __restrict__ d, int N,
                                                                  int C, int raw, float* __restrict__ rec) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = wave_id();
  const int pt0 = blockIdx.x * 32, npx = min(32, N - pt0);
  const bool vec2 = (C & 1) == 0;
  auto pixel_q5 = [&](int n) -> Q5 {   // lane = channel (pair): coalesced rows of G and d
    Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
    if (n >= npx) return q;  // wave-uniform
    const size_t base = ((size_t)b * N + pt0 + n) * C;
    if (vec2) {
      for (int c = lane * 2; c < C; c += 128) {
        const float4 gg = *reinterpret_cast<const float4*>(G + (base + c) * 2);
        const float2 dd = *reinterpret_cast<const float2*>(d + base + c);
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
        const float2 gg = *reinterpret_cast<const float2*>(G + (base + c) * 2);
        const float dd = d[base + c];
        q.m11 = fmaf(gg.x, gg.x, q.m11);
        q.m12 = fmaf(gg.x, gg.y, q.m12);
        q.m22 = fmaf(gg.y, gg.y, q.m22);
        q.g1 = fmaf(gg.x, dd, q.g1);
        q.g2 = fmaf(gg.y, dd, q.g2);
      }
    }
    return q;
  };
  // wave w owns pixels 8w .. 8w+7: transposing butterfly, pixel's sums land on the lanes with (lane & 7) == 0
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
    if (n < npx) {
      const float l11 = sqrtf(fmaxf(q.m11, 0.f));
      const float i11 = l11 > 0.f ? 1.f / l11 : 0.f;
      const float l21 = q.m12 * i11;
      const float l22 = sqrtf(fmaxf(q.m22 - l21 * l21, 0.f));
      const float i22 = l22 > 0.f ? 1.f / l22 : 0.f;
      const float h0 = q.g1 * i11;
      const float h1 = (q.g2 - l21 * h0) * i22;
      float4* o = reinterpret_cast<float4*>(rec + ((size_t)b * N + pt0 + n) * 8);
      if (raw) {   // EquationConstructionGrad wants M and g themselves (eqcon_grad.hip)
        o[0] = make_float4(q.m11, q.m12, q.m22, q.g1);
        o[1] = make_float4(q.g2, 0.f, 0.f, 0.f);
      } else {
        o[0] = make_float4(l11, l21, l22, h0);
        o[1] = make_float4(h1, 0.f, 0.f, 0.f);
      }
    }
  }
}

// ---- W^T W ---------------------------------------------------------------------------------------------
struct EqSyrkArgs {
  const float* J;     // [B][N][2][P]
  const float* rec;   // [B][N][8]
  float* partials;    // [B][Gr][pstride]: P x P then P
  int N, P, Gr, pstride;
  int rb0, cb0;       // first 16-column block of the job's row side / column side
};

// One pass over the pixels = one JOB.  SYM: the upper triangle of the NBC x NBC blocks starting at block cb0 (= rb0) plus
// the record block row (Atb) of those columns -- P <= 144 is the single job <9, 9, true> at 0.  !SYM: all NBR x NBC blocks
// of rows rb0.. x columns cb0.. (an off-diagonal rectangle).  144 < P <= 272 (17 blocks: a wave cannot hold the 153 upper
// blocks) = sym(0, 9) + sym(9, 8) + rect(0..4 x 9..16) + rect(5..8 x 9..16): four passes, every entry of the partial
// written by exactly one job, so the reduction is unchanged.
template <int NBR, int NBC, bool SYM>
__global__ __launch_bounds__(kBlock, 1) void eq_syrk_kernel(const EqSyrkArgs a) {
  constexpr int NPAIR = SYM ? NBC * (NBC + 1) / 2 : NBR * NBC;
  constexpr int NACU = SYM ? NBC : 0, NR = SYM ? 1 : NBR;
  __shared__ float sAcc[NPAIR + NACU][4][64];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id();
  const int N = a.N, P = a.P;
  const int m = lane & 15, kq = lane >> 4;
  const float* __restrict__ J_b = a.J + (size_t)b * N * 2 * P;
  const float* __restrict__ rec_b = a.rec + (size_t)b * N * 8;
  int colc[NBC], rowc[NR];      // this lane's column in block bi, clamped (columns >= P are masked at use)
  bool colok[NBC], rowok[NR];
#pragma unroll
  for (int bi = 0; bi < NBC; ++bi) {
    colok[bi] = 16 * (a.cb0 + bi) + m < P;
    colc[bi] = colok[bi] ? 16 * (a.cb0 + bi) + m : P - 1;
  }
#pragma unroll
  for (int bi = 0; bi < NR; ++bi) {
    rowok[bi] = 16 * (a.rb0 + bi) + m < P;
    rowc[bi] = rowok[bi] ? 16 * (a.rb0 + bi) + m : P - 1;
  }

  f32x4 acc[NPAIR];
  f32x4 acu[NACU ? NACU : 1];
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < (NACU ? NACU : 1); ++q) acu[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's run of 16-pixel steps (32 rows of J); lane (m, kq) holds rows 8 kq .. 8 kq + 7 = pixels 4 kq .. 4 kq + 3
  const int ns = (N + 15) >> 4, nwaves = a.Gr * kNumWaves, gw = g * kNumWaves + w;
  const int s0 = (int)(((long long)ns * gw) / nwaves), s1 = (int)(((long long)ns * (gw + 1)) / nwaves);

  float pj[8][NBC], pjr[8][NR];
  f32x4 pr[4][2];
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t n = (size_t)min(16 * st + 4 * kq + q, N - 1);   // clamped: the prefetch past the last step reads valid memory
      pr[q][0] = *reinterpret_cast<const f32x4*>(rec_b + n * 8);
      pr[q][1] = *reinterpret_cast<const f32x4*>(rec_b + n * 8 + 4);
      const float* r0 = J_b + n * 2 * P;
#pragma unroll
      for (int bi = 0; bi < NBC; ++bi) {
        pj[2 * q][bi] = r0[colc[bi]];
        pj[2 * q + 1][bi] = r0[P + colc[bi]];
      }
      if constexpr (!SYM) {
#pragma unroll
        for (int bi = 0; bi < NR; ++bi) {
          pjr[2 * q][bi] = r0[rowc[bi]];
          pjr[2 * q + 1][bi] = r0[P + rowc[bi]];
        }
      }
    }
  };
  issue(s0);
  for (int st = s0; st < s1; ++st) {
    float l11[4], l21[4], l22[4];
    u32x4_t opu[3];
    {
      float ut[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = 16 * st + 4 * kq + q < N;       // rows of pixels past N: all-zero
        l11[q] = ok ? pr[q][0][0] : 0.f;
        l21[q] = ok ? pr[q][0][1] : 0.f;
        l22[q] = ok ? pr[q][0][2] : 0.f;
        ut[2 * q] = (ok && m == 0) ? pr[q][0][3] : 0.f;  // A-operand row 0 of the record block row = h
        ut[2 * q + 1] = (ok && m == 0) ? pr[q][1][0] : 0.f;
      }
      if constexpr (SYM) split8_bf16x3(ut, opu);
    }
    u32x4_t op[NBC][3], opr[NR][3];
#pragma unroll
    for (int bi = 0; bi < NBC; ++bi) {
      float vv[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float j0 = colok[bi] ? pj[2 * q][bi] : 0.f, j1 = colok[bi] ? pj[2 * q + 1][bi] : 0.f;
        vv[2 * q] = fmaf(l11[q], j0, l21[q] * j1);
        vv[2 * q + 1] = l22[q] * j1;
      }
      split8_bf16x3(vv, op[bi]);
    }
    if constexpr (!SYM) {
#pragma unroll
      for (int bi = 0; bi < NR; ++bi) {
        float vv[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float j0 = rowok[bi] ? pjr[2 * q][bi] : 0.f, j1 = rowok[bi] ? pjr[2 * q + 1][bi] : 0.f;
          vv[2 * q] = fmaf(l11[q], j0, l21[q] * j1);
          vv[2 * q + 1] = l22[q] * j1;
        }
        split8_bf16x3(vv, opr[bi]);
      }
    }
    issue(st + 1);                                          // the raw registers are free again
    __builtin_amdgcn_sched_barrier(0);                      // keep the prefetch ahead of the MFMA block
    if constexpr (SYM) {
#pragma unroll
      for (int bj = 0; bj < NBC; ++bj) acu[bj] = mm6e(opu, op[bj], acu[bj]);
      int idx = 0;
#pragma unroll
      for (int bi = 0; bi < NBC; ++bi)
#pragma unroll
        for (int bj = bi; bj < NBC; ++bj) {
          acc[idx] = mm6e(op[bi], op[bj], acc[idx]);
          ++idx;
        }
    } else {
#pragma unroll
      for (int bi = 0; bi < NR; ++bi)
#pragma unroll
        for (int bj = 0; bj < NBC; ++bj) acc[bi * NBC + bj] = mm6e(opr[bi], op[bj], acc[bi * NBC + bj]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: add the 4 waves in fixed order through LDS, publish the workgroup's partial ----------
  for (int ww = 0; ww < kNumWaves; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int q = 0; q < NPAIR; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) sAcc[q][r][lane] = (ww == 0 ? 0.f : sAcc[q][r][lane]) + acc[q][r];
#pragma unroll
      for (int q = 0; q < NACU; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* c = &sAcc[NPAIR + q][r][lane];
          *c = (ww == 0 ? 0.f : *c) + acu[q][r];
        }
    }
    __syncthreads();
  }
  float* __restrict__ part = a.partials + ((size_t)b * a.Gr + g) * a.pstride;
  const int r = w, brow = 4 * kq + r;   // thread (w, lane) publishes accumulator register r = w: row 4 kq + r, column m
  if constexpr (SYM) {
    int idx = 0;
    for (int bi = 0; bi < NBC; ++bi)
      for (int bj = bi; bj < NBC; ++bj) {
        const int rr = 16 * (a.cb0 + bi) + brow, cc = 16 * (a.cb0 + bj) + m;
        if (rr < P && cc < P && (bj > bi || rr <= cc)) {
          const float v = sAcc[idx][r][lane];
          part[rr * P + cc] = v;
          part[cc * P + rr] = v;
        }
        ++idx;
      }
    if (brow == 0) {
      for (int bj = 0; bj < NBC; ++bj) {
        const int cc = 16 * (a.cb0 + bj) + m;
        if (cc < P) part[P * P + cc] = sAcc[NPAIR + bj][r][lane];
      }
    }
  } else {
    for (int bi = 0; bi < NR; ++bi)
      for (int bj = 0; bj < NBC; ++bj) {
        const int rr = 16 * (a.rb0 + bi) + brow, cc = 16 * (a.cb0 + bj) + m;
        if (rr < P && cc < P) {
          const float v = sAcc[bi * NBC + bj][r][lane];
          part[rr * P + cc] = v;
          part[cc * P + rr] = v;
        }
      }
  }
}

void launch_eq_pixel_records(const float* G, const float* d, int B, int N, int C, int raw, float* rec, hipStream_t s) {
  hipLaunchKernelGGL(eq_pixel_records_kernel, dim3((N + 31) / 32, B), dim3(kBlock), 0, s, G, d, N, C, raw, rec);
}

size_t eq_syrk_record_bytes(int B, int N) { return align_up((size_t)B * N * 8 * sizeof(float), 256); }

int launch_eq_syrk(const float* J, const float* G, const float* d, int B, int N, int C, int P, int nb, int Gr, int pstride,
                   float* partials, float* rec, hipStream_t s) {
  launch_eq_pixel_records(G, d, B, N, C, 0, rec, s);
  EqSyrkArgs a{J, rec, partials, N, P, Gr, pstride, 0, 0};
  const dim3 grid(Gr, B), block(kBlock);
  switch (nb) {
    case 1: hipLaunchKernelGGL((eq_syrk_kernel<1, 1, true>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((eq_syrk_kernel<2, 2, true>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((eq_syrk_kernel<3, 3, true>), grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL((eq_syrk_kernel<5, 5, true>), grid, block, 0, s, a); break;
    case 9: hipLaunchKernelGGL((eq_syrk_kernel<9, 9, true>), grid, block, 0, s, a); break;
    case 17: {   // 144 < P <= 272: four jobs (see eq_syrk_kernel)
      hipLaunchKernelGGL((eq_syrk_kernel<9, 9, true>), grid, block, 0, s, a);
      a.rb0 = a.cb0 = 9;
      hipLaunchKernelGGL((eq_syrk_kernel<8, 8, true>), grid, block, 0, s, a);
      a.rb0 = 0;
      hipLaunchKernelGGL((eq_syrk_kernel<5, 8, false>), grid, block, 0, s, a);
      a.rb0 = 5;
      hipLaunchKernelGGL((eq_syrk_kernel<4, 8, false>), grid, block, 0, s, a);
      break;
    }
    default: return BANET_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
