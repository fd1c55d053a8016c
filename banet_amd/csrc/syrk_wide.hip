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
};

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
template <int PT, bool DD>
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
  for (int st = s0; st < s1; ++st) {
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
      for (int bj = 0; bj < NBV; ++bj)
        part[(size_t)(6 * (a.p0 + pair) + brow % 6) * K + a.koff + 64 * (bj >> 2) + 4 * m + (bj & 3)] = sAcc[NPAIR + j * NBV + bj][r][lane];
    }
  }
  if constexpr (DD) {
    if (brow == 12) {   // Atb_d
#pragma unroll
      for (int bj = 0; bj < NBV; ++bj)
        part[(size_t)6 * P * K + a.koff + 64 * (bj >> 2) + 4 * m + (bj & 3)] = sAcc[NPAIR + bj][r][lane];
    }
    float* pd = part + (size_t)(6 * P + 1) * K;
    int idx = 0;
    for (int bi = 0; bi < NBV; ++bi)
      for (int bj = bi; bj < NBV; ++bj) {
        const int rr = 64 * (bi >> 2) + 4 * brow + (bi & 3), cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
        const float v = sAcc[idx][r][lane];
        if (bj > bi || rr <= cc) {
          pd[(size_t)(a.koff + rr) * K + a.koff + cc] = v;
          pd[(size_t)(a.koff + cc) * K + a.koff + rr] = v;
        }
        ++idx;
      }
  }
}

// ---- rect job: rows r0 .. r0+127, columns c0 .. c0+63 -----------------------------------------------
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
  for (int st = s0; st < s1; ++st) {
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
      const float v = sAcc[bi * NC + bj][r][lane];
      pd[(size_t)rr * K + cc] = v;
      pd[(size_t)cc * K + rr] = v;
    }
}

template <bool DD>
static void launch_slice(const WideArgs& a, int B, int pt, hipStream_t s) {
  const dim3 grid(a.Gs, B), block(kBlock);
  switch (pt) {
    case 1: hipLaunchKernelGGL((ba_syrk_slice_kernel<1, DD>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((ba_syrk_slice_kernel<2, DD>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((ba_syrk_slice_kernel<3, DD>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((ba_syrk_slice_kernel<4, DD>), grid, block, 0, s, a); break;
  }
}

size_t syrk_wide_aux_bytes(int B, int N, int pairs) { return pairs > 1 ? align_up((size_t)B * N * 2 * sizeof(float), 256) : 0; }

int launch_syrk_wide(const float* basis, const float* rec, int B, int N, int K, int pairs, int Gs, int pstride,
                     const int32_t* active, int active_stride, float* partials, float* aux, hipStream_t s) {
  if (K != 128 && K != 256) return BANET_ERR_UNSUPPORTED;
  WideArgs a{basis, rec, nullptr, active, active_stride, partials, N, K, Gs, pstride, pairs, 0, 0, 0, 0};
  if (pairs > 1) {
    hipLaunchKernelGGL(ba_srsum_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, rec, N, pairs, active, active_stride, aux);
    a.srsum = aux;
  }
  for (int koff = 0; koff < K; koff += 128) {
    a.koff = koff;
    for (int p0 = 0; p0 < pairs; p0 += 4) {
      a.p0 = p0;
      const int pt = pairs - p0 < 4 ? pairs - p0 : 4;
      if (p0 == 0)
        launch_slice<true>(a, B, pt, s);
      else
        launch_slice<false>(a, B, pt, s);
    }
  }
  for (int r0 = 0; r0 < K; r0 += 128)        // off-diagonal 128 x 64 rectangles above the diagonal slices
    for (int c0 = r0 + 128; c0 < K; c0 += 64) {
      a.r0 = r0;
      a.c0 = c0;
      hipLaunchKernelGGL(ba_syrk_rect_kernel, dim3(Gs, B), dim3(kBlock), 0, s, a);
    }
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
