// syrk_wide.hip -- the depth blocks of the normal equations for wide bases / many frames on the bf16 matrix pipe:
//   K = 256 (any number of target frames) and K = 128 with more than 4 target frames.
// Same arithmetic as ba_syrk_bf16x6_kernel (syrk.hip): v = sqrt(s) b split exactly into three bf16 pieces, six
// v_mfma_f32_16x16x32_bf16 per 16x16 block and 32 pixels, H_cd / Atb_d through u / sqrt(s).  A wave cannot hold the
// 136 upper blocks of a 256 x 256 matrix (544 accumulator registers), so the matrix is cut into JOBS, each one pass
// over the pixels with the accumulators resident in registers (one wave per SIMD, no LDS, no barrier in the loop):
//   slice(koff, p0, DD)  the symmetric 128 x 128 diagonal slice at coefficient koff (36 blocks, only if DD) plus
//                        H_cd of target frames p0 .. p0+PT-1 (PT <= 4) and, with DD, Atb_d for that slice;
//   rect(r0, c0)         the 128 x 64 off-diagonal rectangle rows r0.., columns c0.. (32 blocks).
// K = 256, 7 target frames: slice(0,0,DD) slice(0,4) slice(128,0,DD) slice(128,4) rect(0,128) rect(0,192) = 3.5 reads
// of the basis, against 7 passes (one per frame) of the LDS-tiled fp32 kernel it replaces (utils.cu:331-414 is the
// reference's materialised form of the same sums).  s and r summed over the window's frames come from a small pre-pass
// (ba_srsum_kernel, 8 bytes per pixel) so that the jobs do not depend on the number of frames.
// Partial layout = the other SYRK kernels' ([B][Gs][(6 pairs + 1) K + K K]): ba_reduce2_kernel is unchanged.
#include "kernels.hpp"
#include "syrk_split.hpp"

namespace banet {

typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));

struct WideArgs {
  const float* basis;  // [B][N][K]
  const float* rec;    // [B][pairs][N][8]
  const float* srsum;  // [B][N][2] (s, r summed over the frames) or nullptr when pairs == 1 (then rec words 6, 7)
  const int32_t* active;
  int active_stride;
  float* partials;
  int N, K, Gs, pstride, pairs;
  int koff, p0;        // slice job
  int r0, c0;          // rect job
  const float* colmax; // F16 (the fp16 two-piece form, see syrk.hip): [B][K] max_n |b_nk|
  const float* recmax; // F16: [B][kWideRecMaxBlocks][2] per-block max s_n, max word^2 / s_n (ba_recmax_kernel, syrk.hip)
};
constexpr int kWideRecMaxBlocks = 32;    // = kRecMaxBlocks of syrk.hip
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));

// fp16 two-piece products, smallest first: lo hi' + hi lo' + hi hi'
__device__ __forceinline__ f32x4 mm3(const u32x4_t (&x)[2], const u32x4_t (&y)[2], f32x4 c) {
  constexpr int kFa[3] = {1, 0, 0}, kFb[3] = {0, 1, 0};
#pragma unroll
  for (int t = 0; t < 3; ++t)
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8w, x[kFa[t]]), __builtin_bit_cast(f16x8w, y[kFb[t]]), c, 0, 0, 0);
  return c;
}

// the window's record maxima -> (2^es > sqrt(max s), 2^eu > sqrt(max word^2 / s)); false if they are not finite
__device__ __forceinline__ bool wide_rec_exponents(const float* recmax, int b, int lane, int& es, int& eu) {
  float smx = 0.f, wmx = 0.f;
  if (lane < kWideRecMaxBlocks) {
    smx = recmax[((size_t)b * kWideRecMaxBlocks + lane) * 2];
    wmx = recmax[((size_t)b * kWideRecMaxBlocks + lane) * 2 + 1];
  }
#pragma unroll
  for (int sh = 1; sh < 64; sh <<= 1) {
    smx = fmaxf(smx, __shfl_xor(smx, sh, 64));
    wmx = fmaxf(wmx, __shfl_xor(wmx, sh, 64));
  }
  (void)frexpf(sqrtf(smx), &es);
  (void)frexpf(sqrtf(wmx), &eu);
  return (smx < __builtin_inff()) && (wmx < __builtin_inff()) && !(smx != smx) && !(wmx != wmx);
}
// column scale 2^(14 - ek - es) for max |b| = cm < 2^ek; ok &= representable
__device__ __forceinline__ float wide_col_scale(float cm, int es, bool& ok, float& inv) {
  int ek = 0;
  (void)frexpf(cm, &ek);
  const int sh = 14 - ek - es;
  ok = ok && (cm < __builtin_inff()) && !(cm != cm) && sh > -100 && sh < 100;
  inv = ldexpf(1.f, -sh);
  return ldexpf(1.f, sh);
}

__global__ __launch_bounds__(256) void ba_srsum_kernel(const float* __restrict__ rec, int N, int pairs,
                                                       const int32_t* active, int active_stride, float* __restrict__ out) {
  const int b = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (active != nullptr && active[(size_t)b * active_stride] == 0) return;
  if (n >= N) return;
  float s = 0.f, r = 0.f;
  for (int p = 0; p < pairs; ++p) {   // fixed order
    const float2 v = *reinterpret_cast<const float2*>(rec + (((size_t)b * pairs + p) * N + n) * 8 + 6);
    s += v.x;
    r += v.y;
  }
  *reinterpret_cast<float2*>(out + ((size_t)b * N + n) * 2) = make_float2(s, r);
}

__device__ __forceinline__ f32x4 mm6(const u32x4_t (&x)[3], const u32x4_t (&y)[3], f32x4 c) {
  constexpr int kTa[6] = {2, 0, 1, 1, 0, 0}, kTb[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
  for (int t = 0; t < 6; ++t)
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, x[kTa[t]]), __builtin_bit_cast(bf16x8w, y[kTb[t]]), c, 0, 0,
                                                0);
  return c;
}

// ---- slice job ---------------------------------------------------------------------------------
template <int PT, bool DD, bool F16>
__global__ __launch_bounds__(kBlock, 1) void ba_syrk_slice_kernel(const WideArgs a) {
  constexpr int KH = 2, NBV = 4 * KH, NPAIR = DD ? NBV * (NBV + 1) / 2 : 0;
  constexpr int NU = (PT + 1) / 2;
  __shared__ float sAcc[NPAIR + NU * NBV][4][64];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int w = wave_id();
  const int N = a.N, K = a.K, P = a.pairs;
  const int m = lane & 15, kq = lane >> 4;
  const float* __restrict__ bas_b = a.basis + (size_t)b * N * K + a.koff;
  const float* __restrict__ rec_b = a.rec + ((size_t)b * P + a.p0) * N * 8;
  // s (and r) of a pixel: summed over the frames by the pre-pass, or the only frame's record
  const float* __restrict__ s_b = a.srsum ? a.srsum + (size_t)b * N * 2 : a.rec + (size_t)b * N * 8 + 6;
  const int sstride = a.srsum ? 2 : 8;
  // record block row j: rows 0-5 / 6-11 = u of frames p0 + 2j / p0 + 2j + 1; with DD row 12 of block row 0 = r (summed)
  const float* ubase[NU];
  int ustride[NU];
  bool uon[NU];
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    uon[j] = false;
    ubase[j] = rec_b;
    ustride[j] = 8;
    if (m < 12) {
      const int pair = 2 * j + m / 6;
      if (pair < PT) {
        uon[j] = true;
        ubase[j] = rec_b + (size_t)pair * N * 8 + m % 6;
      }
    } else if (DD && j == 0 && m == 12) {
      uon[j] = true;
      ubase[j] = s_b + 1;
      ustride[j] = sstride;
    }
  }

  f32x4 acc[DD ? NPAIR : 1];
  f32x4 acu[NU][NBV];
#pragma unroll
  for (int q = 0; q < (DD ? NPAIR : 1); ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NU; ++j)
#pragma unroll
    for (int q = 0; q < NBV; ++q) acu[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ns = (N + 31) >> 5, nwaves = a.Gs * kNumWaves, gw = g * kNumWaves + w;
  const int s0 = (int)(((long long)ns * gw) / nwaves), s1 = (int)(((long long)ns * (gw + 1)) / nwaves);

  // F16: per-column power-of-two scales of this slice's 128 columns + one for the record rows (syrk.hip, T0 = 16)
  [[maybe_unused]] float csc[4 * KH];
  [[maybe_unused]] float usc = 1.f, uinv = 1.f;
  [[maybe_unused]] bool use16 = false;
  [[maybe_unused]] __shared__ float sInv[64 * KH];
  if constexpr (F16) {
    int es = 0, eu = 0;
    bool ok = wide_rec_exponents(a.recmax, b, lane, es, eu);
    usc = ldexpf(1.f, 14 - eu);
    uinv = ldexpf(1.f, eu - 14);
#pragma unroll
    for (int h = 0; h < KH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 64 * h + 4 * m + e;
        float inv;
        csc[4 * h + e] = wide_col_scale(a.colmax[(size_t)b * K + a.koff + col], es, ok, inv);
        if (w == 0 && kq == 0) sInv[col] = inv;
      }
    use16 = __ballot(!ok) == 0ull;
    __syncthreads();
  }

  f32x4 pb[8][KH];
  float ps[8], pu[8][NU];
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t p = (size_t)min(32 * st + 8 * kq + i, N - 1);   // clamped: the prefetch past the last step reads valid memory
#pragma unroll
      for (int h = 0; h < KH; ++h)
        pb[i][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bas_b + p * K + 64 * h + 4 * m));
      ps[i] = s_b[p * sstride];
#pragma unroll
      for (int j = 0; j < NU; ++j) pu[i][j] = ubase[j][p * ustride[j]];
    }
  };
  issue(s0);
  if constexpr (F16) {
    if (use16) {
      for (int st = s0; st < s1; ++st) {
        float sq[8];
        u32x4_t opu[NU][2];
        {
          float ut[NU][8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool ok = 32 * st + 8 * kq + i < N;
            sq[i] = ok ? sqrtf(fmaxf(ps[i], 0.f)) : 0.f;
            const float inv = sq[i] > 0.f ? usc / sq[i] : 0.f;
#pragma unroll
            for (int j = 0; j < NU; ++j) ut[j][i] = uon[j] ? pu[i][j] * inv : 0.f;
          }
#pragma unroll
          for (int j = 0; j < NU; ++j) split8_f16x2(ut[j], opu[j]);
        }
        u32x4_t op[NBV][2];
#pragma unroll
        for (int h = 0; h < KH; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] = (sq[i] * pb[i][h][e]) * csc[4 * h + e];
            split8_f16x2(vv, op[4 * h + e]);
          }
        issue(st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NU; ++j)
#pragma unroll
          for (int bj = 0; bj < NBV; ++bj) acu[j][bj] = mm3(opu[j], op[bj], acu[j][bj]);
        if constexpr (DD) {
          int idx = 0;
#pragma unroll
          for (int bi = 0; bi < NBV; ++bi)
#pragma unroll
            for (int bj = bi; bj < NBV; ++bj) {
              acc[idx] = mm3(op[bi], op[bj], acc[idx]);
              ++idx;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  for (int st = (F16 && use16) ? s1 : s0; st < s1; ++st) {
    float sq[8];
    u32x4_t opu[NU][3];
    {
      float ut[NU][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = 32 * st + 8 * kq + i < N;
        sq[i] = ok ? sqrtf(fmaxf(ps[i], 0.f)) : 0.f;            // zero switches the pixel off
        const float inv = sq[i] > 0.f ? 1.f / sq[i] : 0.f;      // s = 0 implies u = r = 0 (syrk.hip)
#pragma unroll
        for (int j = 0; j < NU; ++j) ut[j][i] = uon[j] ? pu[i][j] * inv : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NU; ++j) split8_bf16x3(ut[j], opu[j]);
    }
    u32x4_t op[NBV][3];
#pragma unroll
    for (int h = 0; h < KH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) vv[i] = sq[i] * pb[i][h][e];
        split8_bf16x3(vv, op[4 * h + e]);
      }
    issue(st + 1);                                          // the raw registers are free again
    __builtin_amdgcn_sched_barrier(0);                      // keep the prefetch ahead of the MFMA block
#pragma unroll
    for (int j = 0; j < NU; ++j)
#pragma unroll
      for (int bj = 0; bj < NBV; ++bj) acu[j][bj] = mm6(opu[j], op[bj], acu[j][bj]);
    if constexpr (DD) {
      int idx = 0;
#pragma unroll
      for (int bi = 0; bi < NBV; ++bi)
#pragma unroll
        for (int bj = bi; bj < NBV; ++bj) {
          acc[idx] = mm6(op[bi], op[bj], acc[idx]);
          ++idx;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: add the 4 waves in fixed order through LDS, un-permute, publish ---------------
  for (int ww = 0; ww < kNumWaves; ++ww) {
    if (w == ww) {
      if constexpr (DD) {
#pragma unroll
        for (int q = 0; q < NPAIR; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) sAcc[q][r][lane] = (ww == 0 ? 0.f : sAcc[q][r][lane]) + acc[q][r];
      }
#pragma unroll
      for (int j = 0; j < NU; ++j)
#pragma unroll
        for (int q = 0; q < NBV; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* c = &sAcc[NPAIR + j * NBV + q][r][lane];
            *c = (ww == 0 ? 0.f : *c) + acu[j][q][r];
          }
    }
    __syncthreads();
  }
  float* __restrict__ part = a.partials + ((size_t)b * a.Gs + g) * a.pstride;
  // thread (w, lane) publishes accumulator register r = w of every block: row 4 kq + r, column m
  const int r = w, brow = 4 * kq + r;
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const int pair = 2 * j + brow / 6;
    if (brow < 12 && pair < PT) {
#pragma unroll
      for (int bj = 0; bj < NBV; ++bj) {
        const int cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
        float v = sAcc[NPAIR + j * NBV + bj][r][lane];
        if constexpr (F16) {
          if (use16) v = (v * uinv) * sInv[cc];
        }
        part[(size_t)(6 * (a.p0 + pair) + brow % 6) * K + a.koff + cc] = v;
      }
    }
  }
  if constexpr (DD) {
    if (brow == 12) {   // Atb_d
#pragma unroll
      for (int bj = 0; bj < NBV; ++bj) {
        const int cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
        float v = sAcc[NPAIR + bj][r][lane];
        if constexpr (F16) {
          if (use16) v = (v * uinv) * sInv[cc];
        }
        part[(size_t)6 * P * K + a.koff + cc] = v;
      }
    }
    float* pd = part + (size_t)(6 * P + 1) * K;
    int idx = 0;
    for (int bi = 0; bi < NBV; ++bi)
      for (int bj = bi; bj < NBV; ++bj) {
        const int rr = 64 * (bi >> 2) + 4 * brow + (bi & 3), cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
        float v = sAcc[idx][r][lane];
        if constexpr (F16) {
          if (use16) v = (v * sInv[rr]) * sInv[cc];
        }
        if (bj > bi || rr <= cc) {
          pd[(size_t)(a.koff + rr) * K + a.koff + cc] = v;
          pd[(size_t)(a.koff + cc) * K + a.koff + rr] = v;
        }
        ++idx;
      }
  }
}

// ---- rect job: rows r0 .. r0+127, columns c0 .. c0+63 -----------------------------------------------
template <bool F16>
__global__ __launch_bounds__(kBlock, 1) void ba_syrk_rect_kernel(const WideArgs a) {
  constexpr int NR = 8, NC = 4;
  __shared__ float sAcc[NR * NC][4][64];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int w = wave_id();
  const int N = a.N, K = a.K, P = a.pairs;
  const int m = lane & 15, kq = lane >> 4;
  const float* __restrict__ row_b = a.basis + (size_t)b * N * K + a.r0;
  const float* __restrict__ col_b = a.basis + (size_t)b * N * K + a.c0;
  const float* __restrict__ s_b = a.srsum ? a.srsum + (size_t)b * N * 2 : a.rec + (size_t)b * N * 8 + 6;
  const int sstride = a.srsum ? 2 : 8;

  f32x4 acc[NR * NC];
#pragma unroll
  for (int q = 0; q < NR * NC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ns = (N + 31) >> 5, nwaves = a.Gs * kNumWaves, gw = g * kNumWaves + w;
  const int s0 = (int)(((long long)ns * gw) / nwaves), s1 = (int)(((long long)ns * (gw + 1)) / nwaves);

  [[maybe_unused]] float cscr[8], cscc[4];
  [[maybe_unused]] bool use16 = false;
  [[maybe_unused]] __shared__ float sInvR[128], sInvC[64];
  if constexpr (F16) {
    int es = 0, eu = 0;
    bool ok = wide_rec_exponents(a.recmax, b, lane, es, eu);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 64 * h + 4 * m + e;
        float inv;
        cscr[4 * h + e] = wide_col_scale(a.colmax[(size_t)b * K + a.r0 + col], es, ok, inv);
        if (w == 0 && kq == 0) sInvR[col] = inv;
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float inv;
      cscc[e] = wide_col_scale(a.colmax[(size_t)b * K + a.c0 + 4 * m + e], es, ok, inv);
      if (w == 0 && kq == 0) sInvC[4 * m + e] = inv;
    }
    use16 = __ballot(!ok) == 0ull;
    __syncthreads();
  }

  f32x4 pr[8][2], pc[8];
  float ps[8];
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t p = (size_t)min(32 * st + 8 * kq + i, N - 1);
      pr[i][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row_b + p * K + 4 * m));
      pr[i][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row_b + p * K + 64 + 4 * m));
      pc[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(col_b + p * K + 4 * m));
      ps[i] = s_b[p * sstride];
    }
  };
  issue(s0);
  if constexpr (F16) {
    if (use16) {
      for (int st = s0; st < s1; ++st) {
        float sq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sq[i] = (32 * st + 8 * kq + i < N) ? sqrtf(fmaxf(ps[i], 0.f)) : 0.f;
        u32x4_t opr[NR][2], opc[NC][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] = (sq[i] * pr[i][h][e]) * cscr[4 * h + e];
            split8_f16x2(vv, opr[4 * h + e]);
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float vv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) vv[i] = (sq[i] * pc[i][e]) * cscc[e];
          split8_f16x2(vv, opc[e]);
        }
        issue(st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bi = 0; bi < NR; ++bi)
#pragma unroll
          for (int bj = 0; bj < NC; ++bj) acc[bi * NC + bj] = mm3(opr[bi], opc[bj], acc[bi * NC + bj]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  for (int st = (F16 && use16) ? s1 : s0; st < s1; ++st) {
    float sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sq[i] = (32 * st + 8 * kq + i < N) ? sqrtf(fmaxf(ps[i], 0.f)) : 0.f;
    u32x4_t opr[NR][3], opc[NC][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) vv[i] = sq[i] * pr[i][h][e];
        split8_bf16x3(vv, opr[4 * h + e]);
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float vv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) vv[i] = sq[i] * pc[i][e];
      split8_bf16x3(vv, opc[e]);
    }
    issue(st + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bi = 0; bi < NR; ++bi)
#pragma unroll
      for (int bj = 0; bj < NC; ++bj) acc[bi * NC + bj] = mm6(opr[bi], opc[bj], acc[bi * NC + bj]);
    __builtin_amdgcn_sched_barrier(0);
  }

  for (int ww = 0; ww < kNumWaves; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int q = 0; q < NR * NC; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) sAcc[q][r][lane] = (ww == 0 ? 0.f : sAcc[q][r][lane]) + acc[q][r];
    }
    __syncthreads();
  }
  float* pd = a.partials + ((size_t)b * a.Gs + g) * a.pstride + (size_t)(6 * P + 1) * K;
  const int r = w, brow = 4 * kq + r;
  for (int bi = 0; bi < NR; ++bi)
    for (int bj = 0; bj < NC; ++bj) {
      const int rr = a.r0 + 64 * (bi >> 2) + 4 * brow + (bi & 3), cc = a.c0 + 4 * m + bj;
      float v = sAcc[bi * NC + bj][r][lane];
      if constexpr (F16) {
        if (use16) v = (v * sInvR[rr - a.r0]) * sInvC[cc - a.c0];
      }
      pd[(size_t)rr * K + cc] = v;
      pd[(size_t)cc * K + rr] = v;
    }
}

template <bool DD, bool F16>
static void launch_slice(const WideArgs& a, int B, int pt, hipStream_t s) {
  const dim3 grid(a.Gs, B), block(kBlock);
  switch (pt) {
    case 1: hipLaunchKernelGGL((ba_syrk_slice_kernel<1, DD, F16>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((ba_syrk_slice_kernel<2, DD, F16>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((ba_syrk_slice_kernel<3, DD, F16>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((ba_syrk_slice_kernel<4, DD, F16>), grid, block, 0, s, a); break;
  }
}

size_t syrk_wide_aux_bytes(int B, int N, int pairs) { return pairs > 1 ? align_up((size_t)B * N * 2 * sizeof(float), 256) : 0; }

int launch_syrk_wide(const float* basis, const float* rec, int B, int N, int K, int pairs, int Gs, int pstride,
                     const int32_t* active, int active_stride, float* partials, float* aux, hipStream_t s, const float* colmax,
                     const float* recmax) {
  if (K != 128 && K != 256) return BANET_ERR_UNSUPPORTED;
  const bool f16 = colmax != nullptr && recmax != nullptr;       // the fp16 two-piece form (scales prepared by launch_syrk)
  WideArgs a{basis, rec, nullptr, active, active_stride, partials, N, K, Gs, pstride, pairs, 0, 0, 0, 0, colmax, recmax};
  if (pairs > 1) {
    hipLaunchKernelGGL(ba_srsum_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, rec, N, pairs, active, active_stride, aux);
    a.srsum = aux;
  }
  for (int koff = 0; koff < K; koff += 128) {
    a.koff = koff;
    for (int p0 = 0; p0 < pairs; p0 += 4) {
      a.p0 = p0;
      const int pt = pairs - p0 < 4 ? pairs - p0 : 4;
      if (p0 == 0) {
        if (f16) launch_slice<true, true>(a, B, pt, s);
        else launch_slice<true, false>(a, B, pt, s);
      } else {
        if (f16) launch_slice<false, true>(a, B, pt, s);
        else launch_slice<false, false>(a, B, pt, s);
      }
    }
  }
  for (int r0 = 0; r0 < K; r0 += 128)        // off-diagonal 128 x 64 rectangles above the diagonal slices
    for (int c0 = r0 + 128; c0 < K; c0 += 64) {
      a.r0 = r0;
      a.c0 = c0;
      if (f16) hipLaunchKernelGGL(ba_syrk_rect_kernel<true>, dim3(Gs, B), dim3(kBlock), 0, s, a);
      else hipLaunchKernelGGL(ba_syrk_rect_kernel<false>, dim3(Gs, B), dim3(kBlock), 0, s, a);
    }
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
