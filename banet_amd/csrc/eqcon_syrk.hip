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
#include <cstdlib>

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

// ---- 144 < P <= 272 in ONE pass over J -----------------------------------------------------------------------------------
// The 17 x 17 block triangle (+ the record block row) does not fit one wave's accumulators, so it is cut into four jobs:
// sym(0..8), sym(9..16), rect(0..4 x 9..16), rect(5..8 x 9..16).  Run as four launches each job reads J again and splits the
// columns it needs again (42 column-block splits per step for 17 distinct ones) and the op runs at the fabric's rate on 2.7 x its
// bytes (0.23 of the roofline: profiles/r02_run9_eqcon_op.txt).  Here the four jobs are the four WAVES of a workgroup (one per
// SIMD, up to 512 registers each) working on the same 16 pixels per step:
//   split   every wave takes a few of the 18 operand blocks (17 column blocks + the record row): loads its columns of J one
//           step ahead into registers, weights (w = L^T J), splits and writes the bf16 triples to LDS (3 KB per block, two
//           buffers) -- each block is loaded and split ONCE per step, the jobs with fewer tiles take more blocks;
//   MFMA    every wave reads the operand blocks of its job into registers (42 x 3 KB per step and workgroup: ~1 k LDS cycles
//           against ~5 k matrix-pipe cycles) and runs its tiles; accumulators stay in registers for the whole range and are
//           published straight from them (every entry of the partial is owned by exactly one job).
// One barrier per step: [barrier] operands(st) -> registers, split(st + 1) -> the other buffer, request rows(st + 2), MFMAs(st).
#ifndef BANET_EQ4_EXP
#define BANET_EQ4_EXP 0
#endif
constexpr int kEq4Blocks = 18;                                      // 17 column blocks of J + the record block row
constexpr int kEq4OpVecs = kEq4Blocks * 3 * 64;                     // one step's operands: [block][piece][lane] of 16 B
constexpr int kEq4LdsBytes = 2 * kEq4OpVecs * 16;

// Job = the tiles one wave owns: SYM: the upper triangle of column blocks cb0 .. cb0 + NBC - 1; !SYM: the rectangle of row blocks
// rb0 .. x column blocks cb0 ..; plus NU tiles of the record block row (Atb): UKIND 1 = over the job's column blocks U0 .. U0 + NU - 1,
// UKIND 2 = over its ROW blocks (a rectangle holds those as operands too).  The wave also splits operand blocks S0 .. S0 + NS - 1.
template <int NBR, int NBC, bool SYM, int S0, int NS, int UKIND, int U0, int NU>
__device__ __forceinline__ void eq_syrk4_job(const EqSyrkArgs& a, int rb0, int cb0, u32x4_t* __restrict__ sOp) {
  constexpr int NPAIR = SYM ? NBC * (NBC + 1) / 2 : NBR * NBC;
  constexpr int NR = SYM ? 0 : NBR;
  constexpr int kRec = 17;
  static_assert(UKIND == 0 || (UKIND == 1 && U0 + NU <= NBC) || (UKIND == 2 && !SYM && U0 + NU <= NBR), "record tiles out of range");
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63;
  const int N = a.N, P = a.P;
  const int m = lane & 15, kq = lane >> 4;
  const float* __restrict__ J_b = a.J + (size_t)b * N * 2 * P;
  const float* __restrict__ rec_b = a.rec + (size_t)b * N * 8;
  // this lane's column of every block it splits; columns >= P take column 0: their operands are finite and only reach entries
  // of the partial that are never published (a tile entry depends on one column of each of its two blocks)
  int colc[NS];
#pragma unroll
  for (int sb = 0; sb < NS; ++sb) colc[sb] = ((S0 + sb) < kRec && 16 * (S0 + sb) + m < P) ? 16 * (S0 + sb) + m : 0;
  f32x4 acc[NPAIR];
  f32x4 acu[NU ? NU : 1];
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < (NU ? NU : 1); ++q) acu[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ns = (N + 15) >> 4;   // 16-pixel steps; the workgroup's range, the same for its four waves
  const int s0 = (int)(((long long)ns * g) / a.Gr), s1 = (int)(((long long)ns * (g + 1)) / a.Gr);
  struct Rec5 {   // the five used floats of a pixel's record
    f32x4 a;
    float h1;
  };
  float pj[8][NS];
  Rec5 pr[4];
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t n = (size_t)min(16 * st + 4 * kq + q, N - 1);   // clamped: the prefetch past the last step reads valid memory
      pr[q].a = *reinterpret_cast<const f32x4*>(rec_b + n * 8);
      pr[q].h1 = rec_b[n * 8 + 4];
      const float* r0 = J_b + n * 2 * P;
#pragma unroll
      for (int sb = 0; sb < NS; ++sb) {
        pj[2 * q][sb] = r0[colc[sb]];
        pj[2 * q + 1][sb] = r0[P + colc[sb]];
      }
    }
  };
  auto split = [&](int st) __attribute__((always_inline)) {       // the prefetched rows of step st -> operand buffer (st - s0) & 1
    u32x4_t* __restrict__ buf = sOp + (size_t)((st - s0) & 1) * kEq4OpVecs;
    float l11[4], l21[4], l22[4], ut[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = 16 * st + 4 * kq + q < N;       // rows of pixels past N: all-zero
      l11[q] = ok ? pr[q].a[0] : 0.f;
      l21[q] = ok ? pr[q].a[1] : 0.f;
      l22[q] = ok ? pr[q].a[2] : 0.f;
      ut[2 * q] = (ok && m == 0) ? pr[q].a[3] : 0.f;  // A-operand row 0 of the record block row = h
      ut[2 * q + 1] = (ok && m == 0) ? pr[q].h1 : 0.f;
    }
#pragma unroll
    for (int sb = 0; sb < NS; ++sb) {
      float vv[8];
      if (S0 + sb == kRec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = ut[e];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vv[2 * q] = fmaf(l11[q], pj[2 * q][sb], l21[q] * pj[2 * q + 1][sb]);
          vv[2 * q + 1] = l22[q] * pj[2 * q + 1][sb];
        }
      }
      u32x4_t t[3];
#if BANET_EQ4_EXP == 2
      t[0] = t[1] = t[2] = u32x4_t{(unsigned)__float_as_uint(vv[0]), 0u, 0u, 0u};
#else
      split8_bf16x3(vv, t);
#endif
#pragma unroll
      for (int p = 0; p < 3; ++p) buf[((S0 + sb) * 3 + p) * 64 + lane] = t[p];
    }
  };
  issue(s0);
  split(s0);
  issue(s0 + 1);
  for (int st = s0; st < s1; ++st) {
    const u32x4_t* __restrict__ buf = sOp + (size_t)((st - s0) & 1) * kEq4OpVecs;
    // step st's operands are complete; every wave has finished reading the other buffer (step st - 1's)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    u32x4_t op[NBC][3], opr[NR ? NR : 1][3], opu[3];
#pragma unroll
    for (int bi = 0; bi < NBC; ++bi)
#pragma unroll
      for (int p = 0; p < 3; ++p) op[bi][p] = buf[((cb0 + bi) * 3 + p) * 64 + lane];
#pragma unroll
    for (int bi = 0; bi < NR; ++bi)
#pragma unroll
      for (int p = 0; p < 3; ++p) opr[bi][p] = buf[((rb0 + bi) * 3 + p) * 64 + lane];
    if constexpr (NU > 0) {
#pragma unroll
      for (int p = 0; p < 3; ++p) opu[p] = buf[(kRec * 3 + p) * 64 + lane];
    }
    // the split of step st + 1 FIRST (its VALU work covers the latency of the operand reads above), then the request for the rows
    // of step st + 2: they have the whole MFMA phase to arrive.  (A second register set to request them a block earlier does not
    // fit beside 54 accumulator tiles; interleaving the split into the MFMA stream by sched_group_barrier: +3 %, spills with it.)
    split(st + 1);      // past the range: the clamped prefetch, written to the other buffer and never read
    issue(st + 2);
    __builtin_amdgcn_sched_barrier(0);
#if BANET_EQ4_EXP == 1
#pragma unroll
    for (int bi = 0; bi < NBC; ++bi) acc[bi][0] += __uint_as_float(op[bi][0][0] ^ op[bi][1][1] ^ op[bi][2][2]);
    if (false)
#endif
    {
#pragma unroll
      for (int j = 0; j < NU; ++j) acu[j] = mm6e(opu, UKIND == 2 ? opr[U0 + j] : op[U0 + j], acu[j]);
      if constexpr (SYM) {
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
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- publish: accumulator layout = lane (m, kq) holds rows 4 kq + v (v = 0..3) of column m
  float* __restrict__ part = a.partials + ((size_t)b * a.Gr + g) * a.pstride;
  if (kq == 0) {
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int cc = 16 * ((UKIND == 2 ? rb0 : cb0) + U0 + j) + m;
      if (cc < P) part[P * P + cc] = acu[j][0];
    }
  }
  if constexpr (SYM) {
    int idx = 0;
#pragma unroll
    for (int bi = 0; bi < NBC; ++bi)
#pragma unroll
      for (int bj = bi; bj < NBC; ++bj) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rr = 16 * (cb0 + bi) + 4 * kq + v, cc = 16 * (cb0 + bj) + m;
          if (rr < P && cc < P && (bj > bi || rr <= cc)) {
            part[rr * P + cc] = acc[idx][v];
            part[cc * P + rr] = acc[idx][v];
          }
        }
        ++idx;
      }
  } else {
#pragma unroll
    for (int bi = 0; bi < NR; ++bi)
#pragma unroll
      for (int bj = 0; bj < NBC; ++bj)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int rr = 16 * (rb0 + bi) + 4 * kq + v, cc = 16 * (cb0 + bj) + m;
          if (rr < P && cc < P) {
            part[rr * P + cc] = acc[bi * NBC + bj][v];
            part[cc * P + rr] = acc[bi * NBC + bj][v];
          }
        }
  }
}

__global__ __launch_bounds__(kBlock, 1) void eq_syrk4_kernel(const EqSyrkArgs a) {
  extern __shared__ u32x4_t sEq4[];
  const int w = wave_id();
  // tiles per wave 49 / 36 / 45 / 40 (the 17 record tiles are spread so that no wave has more than 49 of the 170), operand
  // blocks split per wave 2 / 6 / 4 / 6: a tile costs 6 MFMAs = ~100 cycles, a block ~75 instructions of a single wave = ~300
  if (w == 0)
    eq_syrk4_job<9, 9, true, 0, 2, 1, 5, 4>(a, 0, 0, sEq4);        // sym(0..8) + record tiles of blocks 5..8
  else if (w == 1)
    eq_syrk4_job<8, 8, true, 2, 6, 0, 0, 0>(a, 9, 9, sEq4);        // sym(9..16)
  else if (w == 2)
    eq_syrk4_job<5, 8, false, 8, 4, 2, 0, 5>(a, 0, 9, sEq4);       // rect(0..4 x 9..16) + record tiles of blocks 0..4
  else
    eq_syrk4_job<4, 8, false, 12, 6, 1, 0, 8>(a, 5, 9, sEq4);      // rect(5..8 x 9..16) + record tiles of blocks 9..16
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
    case 17: {   // 144 < P <= 272: four jobs, side by side in one launch (BANET_EQ_FOUR_PASS=1: one launch each, for A/B)
      static const bool four_pass = getenv("BANET_EQ_FOUR_PASS") != nullptr && getenv("BANET_EQ_FOUR_PASS")[0] == '1';
      if (!four_pass) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_syrk4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kEq4LdsBytes);
        hipLaunchKernelGGL(eq_syrk4_kernel, grid, block, kEq4LdsBytes, s, a);
        break;
      }
      hipLaunchKernelGGL((eq_syrk_kernel<9, 9, true>), grid, block, 0, s, a);
      a.rb0 = a.cb0 = 9;
      hipLaunchKernelGGL((eq_syrk_kernel<8, 8, true>), grid, block, 0, s, a);
      a.rb0 = 0;
      hipLaunchKernelGGL((eq_syrk_kernel<5, 8, false>), grid, block, 0, s, a);
      a.rb0 = 5;
      hipLaunchKernelGGL((eq_syrk_kernel<4, 8, false>), grid, block, 0, s, a);
      break;
    }
    case 19: {   // 272 < P <= 304 (round 5; cfg-5's 8-frame windows: P = 298): the 17 x 17 block part in ONE pass over J by the four
                 // wave-jobs above (all its columns are < 272 <= P), then the two extra column blocks 17, 18 as three small jobs of the
                 // generic job kernel: sym(17..18) with their record tiles, rect(0..8 x 17..18), rect(9..16 x 17..18) -- they read
                 // 2 + 11 + 10 of the 19 column blocks again (1.2 more passes over J); every entry of the partial has one owner
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_syrk4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kEq4LdsBytes);
      hipLaunchKernelGGL(eq_syrk4_kernel, grid, block, kEq4LdsBytes, s, a);
      a.rb0 = a.cb0 = 17;
      hipLaunchKernelGGL((eq_syrk_kernel<2, 2, true>), grid, block, 0, s, a);
      a.rb0 = 0;
      hipLaunchKernelGGL((eq_syrk_kernel<9, 2, false>), grid, block, 0, s, a);
      a.rb0 = 9;
      hipLaunchKernelGGL((eq_syrk_kernel<8, 2, false>), grid, block, 0, s, a);
      break;
    }
    default: return BANET_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
