// The depth-basis blocks of the normal equations on gfx950.  Three kernels compute the same sums:
//   ba_syrk_bf16x6_kernel  the product path for K = 64 / 128 and <= 4 target frames: bf16 matrix pipe at fp32 accuracy,
//                          operands straight from registers (see its header comment further down);
//                          K = 256 and more frames: the same scheme cut into jobs, syrk_wide.hip;
//   ba_syrk_direct_kernel  its fp32-MFMA predecessor, kept for A/B (flags bit 8);
//   ba_syrk_kernel<NB>     LDS-tiled fp32 MFMA, any K <= 256 (the remaining basis sizes), described first:
//
// ba_syrk_kernel -- the depth-basis blocks of the normal equations on gfx950:
//   H_dd = sum_n s_n b_n b_n^T   (K x K, fp32 MFMA v_mfma_f32_16x16x4_f32, upper 16x16 blocks)
//   H_cd = sum_n u_n b_n^T       (6 x K, VALU rank-1 updates)
//   Atb_d = sum_n r_n b_n        (K)
// from the basis [N,K] (streamed once, 64-pixel tiles = 32 KB contiguous at K=128) and the
// per-pixel records (u,s,r) written by ba_gather_kernel.  This is the part of
// EquationConstruction (utils.cu:331-414) that involves J's depth columns jd (x) b_n
// (bundlenet.py:260-261), never materialised here.
// Pipeline: registers <- tile t+1 (global loads issued before the MFMA phase of tile t),
// LDS double buffer, one barrier per tile.  Wave w owns block rows w and NB-1-w of the upper
// triangle (NB+1 blocks: balanced); accumulators stay in registers for the whole kernel.
#include <algorithm>
#include "kernels.hpp"
#include "mlp.hpp"
#include "syrk_split.hpp"

namespace banet {

struct SyrkArgs {
  const float* basis;  // [B][N][K]
  const float* rec;    // [B][pairs][N][8]
  const int32_t* active;
  int active_stride;
  float* partials;     // [B][Gs][(6 pairs + 1) K + K*K]: H_cd rows (pair, c), the Atb_d row, H_dd
  int N, K, Gs, tiles, pstride;
  int pairs;           // target frames per window (records of pair i: rec + ((b pairs + i) N) 8)
  int pass;            // LDS-tiled kernel only: pass p adds H_cd of pair p; pass 0 also H_dd / Atb_d with s, r summed over pairs
  MlpRole mr;          // ba_syrk_bf16x6_kernel: workgroup Gs of every window evaluates the lambda MLP (mr.y != nullptr)
  const float* colmax; // T0 = 16 (fp16 two-piece split): [B][K] max_n |b_nk| (ba_colmax_kernel)
  const float* recmax; // T0 = 16: [B][kRecMaxBlocks][2] per-block max_n s_n and max_n,word word^2 / s_n (ba_recmax_kernel)
};
constexpr int kRecMaxBlocks = 32;

template <int NB>
__global__ __launch_bounds__(kBlock, NB > 8 ? 1 : 2) void ba_syrk_kernel(const SyrkArgs a) {
  constexpr int PW = NB > 8 ? 2 : 1;    // block-row pairs per wave (NB = 16, K <= 256: rows {w, 15-w} and {w+4, 11-w})
  constexpr int KPAD = NB * 16;
  constexpr int KP = KPAD + 16;         // row stride: bank shift 16 per row -> conflict-free MFMA operand reads
  constexpr int KV = KPAD >= 64 ? KPAD / 64 : 1;
  constexpr int QPR = KPAD / 4;         // float4 per row
  constexpr int QT = (kTilePix * QPR + kBlock - 1) / kBlock;  // float4 per thread per tile
  constexpr int NSLOT = NB + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sB0 = smem;                          // [2][64][KP]
  float* sU0 = smem + 2 * kTilePix * KP;      // [2][64][8]
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int w = wave_id();
  const int N = a.N, K = a.K;
  const float* __restrict__ bas_b = a.basis + (size_t)b * N * K;
  const float* __restrict__ rec_b = a.rec + ((size_t)b * a.pairs + a.pass) * N * 8;   // this pass's pair
  const bool k4 = (K & 3) == 0;

  float hcd[7][KV];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int e = 0; e < KV; ++e) hcd[i][e] = 0.f;
  f32x4 acc[PW][NSLOT];
#pragma unroll
  for (int p = 0; p < PW; ++p)
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int npairs = (NB + 1) / 2;
  const bool mf_on = w < npairs && a.pass == 0;

  float4 pre[QT];
  float4 preu = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_tile = [&](int t) {
    const int pt0 = t * kTilePix;
    if (k4) {  // uniform; kept OUTSIDE the unrolled loop and branch-free inside it, so that all row
               // loads of the tile are in flight together (an `if` per load serialises them on vmcnt(0))
#pragma unroll
      for (int i = 0; i < QT; ++i) {
        const int idx = tid + i * kBlock;
        const int n = idx / QPR, q = idx - n * QPR;
        const bool ok = idx < kTilePix * QPR && pt0 + n < N && 4 * q < K;
        const float4 v = *reinterpret_cast<const float4*>(bas_b + (ok ? (size_t)(pt0 + n) * K + 4 * q : 0));
        pre[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < QT; ++i) {
        const int idx = tid + i * kBlock;
        const int n = idx / QPR, q = idx - n * QPR;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < kTilePix * QPR && pt0 + n < N) {
          const float* p = bas_b + (size_t)(pt0 + n) * K + 4 * q;
          if (4 * q + 0 < K) v.x = p[0];
          if (4 * q + 1 < K) v.y = p[1];
          if (4 * q + 2 < K) v.z = p[2];
          if (4 * q + 3 < K) v.w = p[3];
        }
        pre[i] = v;
      }
    }
    {
      const int n = tid >> 1;
      const bool ok = tid < 2 * kTilePix && pt0 + n < N;
      float4 v = *reinterpret_cast<const float4*>(rec_b + (ok ? (size_t)(pt0 + n) * 8 + 4 * (tid & 1) : 0));
      if (a.pass > 0) {   // H_cd of this pair only: no s (MFMA phase is off), no r
        if (tid & 1) v.w = 0.f;
      } else if ((tid & 1) && ok) {   // pass 0 of a multi-frame window: s and r summed over the pairs
        for (int i = 1; i < a.pairs; ++i) {
          const float4 o = *reinterpret_cast<const float4*>(rec_b + ((size_t)i * N + pt0 + n) * 8 + 4);
          v.z += o.z;
          v.w += o.w;
        }
      }
      preu = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int buf) {
    float* sB = sB0 + buf * kTilePix * KP;
    float* sU = sU0 + buf * kTilePix * kUStrideS;
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      const int idx = tid + i * kBlock;
      const int n = idx / QPR, q = idx - n * QPR;
      if (idx < kTilePix * QPR) *reinterpret_cast<float4*>(sB + n * KP + 4 * q) = pre[i];
    }
    if (tid < 2 * kTilePix) *reinterpret_cast<float4*>(sU + (tid >> 1) * kUStrideS + 4 * (tid & 1)) = preu;
  };

  int t = g;
  int cur = 0;
  if (t < a.tiles) {
    load_tile(t);
    store_tile(0);
  }
  __syncthreads();
  for (; t < a.tiles; t += a.Gs) {
    const bool more = t + a.Gs < a.tiles;
    if (more) load_tile(t + a.Gs);          // global loads in flight during the compute below
    const float* sB = sB0 + cur * kTilePix * KP;
    const float* sU = sU0 + cur * kTilePix * kUStrideS;
    // ---- H_cd += u_n b_n^T, Atb_d += r_n b_n ;  lane = coefficient(s), wave w pixels 16w.. ----
    {
      const int kb = lane * KV;
      if (kb < KPAD) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
          const int n = 16 * w + i;
          const float4 ua = *reinterpret_cast<const float4*>(sU + n * kUStrideS);
          const float4 ub = *reinterpret_cast<const float4*>(sU + n * kUStrideS + 4);
#pragma unroll
          for (int e = 0; e < KV; ++e) {
            const float bv = sB[n * KP + kb + e];
            hcd[0][e] = fmaf(ua.x, bv, hcd[0][e]);
            hcd[1][e] = fmaf(ua.y, bv, hcd[1][e]);
            hcd[2][e] = fmaf(ua.z, bv, hcd[2][e]);
            hcd[3][e] = fmaf(ua.w, bv, hcd[3][e]);
            hcd[4][e] = fmaf(ub.x, bv, hcd[4][e]);
            hcd[5][e] = fmaf(ub.y, bv, hcd[5][e]);
            hcd[6][e] = fmaf(ub.w, bv, hcd[6][e]);
          }
        }
      }
    }
    // ---- H_dd += sum_n s_n b_n b_n^T on the matrix cores ---------------------------------
    if (mf_on) {
      const int col = lane & 15, kq = lane >> 4;
#pragma unroll 2
      for (int kk = 0; kk < kTilePix / 4; ++kk) {
        const int pix = 4 * kk + kq;
        const float* row = sB + pix * KP + col;
        const float sv = sU[pix * kUStrideS + 6];
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          const int i1 = w + 4 * p, i2 = NB - 1 - i1;
          if (i1 > i2) continue;  // wave-uniform (odd NB / fewer pairs than waves)
          const int n1 = NB - i1;
          const int nslots = (i2 != i1) ? NB + 1 : n1;
          const float a1 = sv * row[16 * i1];
          const float a2 = sv * row[16 * i2];
#pragma unroll
          for (int q = 0; q < NSLOT; ++q) {
            if (q < nslots) {  // wave-uniform
              const bool first = q < n1;
              const float bj = row[16 * (first ? i1 + q : i2 + q - n1)];
              acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(first ? a1 : a2, bj, acc[p][q], 0, 0, 0);
            }
          }
        }
      }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue ------------------------------------------------------------------------
  float* __restrict__ part = a.partials + ((size_t)b * a.Gs + g) * a.pstride;
  float* sH = smem;  // [4][8][KPAD] overlays the tile buffers (all waves are past the last barrier)
  {
    const int kb = lane * KV;
    if (kb < KPAD) {
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < KV; ++e) sH[(w * 8 + i) * KPAD + kb + e] = hcd[i][e];
    }
  }
  __syncthreads();
  for (int e = tid; e < 7 * K; e += kBlock) {
    const int i = e / K, k = e - i * K;
    if (i == 6 && a.pass > 0) continue;   // Atb_d comes from pass 0
    const int row = i < 6 ? 6 * a.pass + i : 6 * a.pairs;
    part[row * K + k] = (sH[(0 * 8 + i) * KPAD + k] + sH[(1 * 8 + i) * KPAD + k]) +
                        (sH[(2 * 8 + i) * KPAD + k] + sH[(3 * 8 + i) * KPAD + k]);
  }
  if (mf_on) {
    float* pd = part + (6 * a.pairs + 1) * K;
    const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int p = 0; p < PW; ++p) {
      const int i1 = w + 4 * p, i2 = NB - 1 - i1;
      if (i1 > i2) continue;
      const int n1 = NB - i1;
      const int nslots = (i2 != i1) ? NB + 1 : n1;
#pragma unroll
      for (int q = 0; q < NSLOT; ++q) {
        if (q < nslots) {
          const bool first = q < n1;
          const int bi = first ? i1 : i2, bj = first ? i1 + q : i2 + q - n1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * bi + rq + r, cc = 16 * bj + col;
            if (rr < K && cc < K && (bj > bi || rr <= cc)) {
              pd[rr * K + cc] = acc[p][q][r];
              pd[cc * K + rr] = acc[p][q][r];
            }
          }
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------
// ba_syrk_direct_kernel -- K = 64 KH (the reference's K = 128 is KH = 2): the same sums without LDS
// and without workgroup barriers.  A wave owns a contiguous run of pixel quads and ALL 16x16
// blocks of the upper triangle (36 for K = 128: 144 accumulator registers).  Lane (m, kq) of a
// quad loads pixel 4q + kq's coefficients {4m..4m+3} + 64h as 16-byte loads (a wave instruction =
// 4 x 256 contiguous bytes) and uses every loaded value as the A operand (scaled by s_n) of one
// "virtual" block row and the B operand of one virtual block column: virtual block i' holds the
// coefficients 64 (i' >> 2) + 4 m + (i' & 3), m = 0..15 -- a permutation that is undone when the
// partial is written.  Loads run one batch of quads ahead of the matrix cores (register double
// buffer); H_cd / Atb_d are packed-fp32 FMAs issued in the shadow of the MFMAs.
// --------------------------------------------------------------------------------------
constexpr int kSyrkBatch = 4;

template <int KH, int PAIRS>
__global__ __launch_bounds__(kBlock, 1) void ba_syrk_direct_kernel(const SyrkArgs a) {
  constexpr int NBV = 4 * KH, NPAIR = NBV * (NBV + 1) / 2, K = 64 * KH, NV = 4 * KH;
  constexpr int NU = (PAIRS + 1) / 2;   // record block rows: rows 0..5 / 6..11 = u of pairs 2j / 2j+1; block row 0 rows 12+i = r_i
  __shared__ float sAcc[NPAIR + NU * NBV][4][64];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int w = wave_id();
  const int N = a.N;
  const int m = lane & 15, kq = lane >> 4;
  const float* __restrict__ bas_b = a.basis + (size_t)b * N * K;
  const float* __restrict__ rec_b = a.rec + (size_t)b * PAIRS * N * 8;
  // record word this lane feeds to block row j: (pair, word) or none
  size_t uoff[NU];
  bool uon[NU];
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    int pair = -1, word = 0;
    if (m < 12) {
      pair = 2 * j + m / 6;
      word = m % 6;
    } else if (j == 0) {
      pair = m - 12;
      word = 7;
    }
    uon[j] = pair >= 0 && pair < PAIRS;
    uoff[j] = uon[j] ? (size_t)pair * N * 8 + word : 0;
  }

  f32x4 acc[NPAIR];      // H_dd, upper triangle of virtual blocks
  f32x4 acu[NU][NBV];    // record block rows against every virtual block column
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NU; ++j)
#pragma unroll
    for (int q = 0; q < NBV; ++q) acu[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's run of quads
  const int nq = (N + 3) >> 2, nwaves = a.Gs * kNumWaves, gw = g * kNumWaves + w;
  const int q0 = (int)(((long long)nq * gw) / nwaves), q1 = (int)(((long long)nq * (gw + 1)) / nwaves);

  // register double buffer, kSyrkBatch quads per batch: the loads of batch t+1 are issued (and pinned
  // with a scheduling barrier) before batch t is multiplied, so they have a whole batch of MFMAs
  // (kSyrkBatch x 44 x 32 cycles) to arrive
  struct Batch {
    f32x4 pb[kSyrkBatch][KH];
    float ps[kSyrkBatch][PAIRS], pu[kSyrkBatch][NU];
  };
  auto issue = [&](Batch& B_, int q) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < kSyrkBatch; ++d) {
      const int pix = 4 * (q + d) + kq;
      const size_t p = (q + d < q1 && pix < N) ? (size_t)pix : 0;
#pragma unroll
      for (int h = 0; h < KH; ++h)
        B_.pb[d][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bas_b + p * K + 64 * h + 4 * m));
#pragma unroll
      for (int i = 0; i < PAIRS; ++i) B_.ps[d][i] = rec_b[((size_t)i * N + p) * 8 + 6];
#pragma unroll
      for (int j = 0; j < NU; ++j) B_.pu[d][j] = rec_b[p * 8 + uoff[j]];
    }
  };
  auto consume = [&](const Batch& B_, int q) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < kSyrkBatch; ++d) {
      const bool ok = q + d < q1 && 4 * (q + d) + kq < N;
      float ssum = B_.ps[d][0];
#pragma unroll
      for (int i = 1; i < PAIRS; ++i) ssum += B_.ps[d][i];
      const float sv = ok ? ssum : 0.f;                     // zero records switch the pixel off
      float bv[NV], av[NV];
#pragma unroll
      for (int h = 0; h < KH; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bv[4 * h + e] = B_.pb[d][h][e];
          av[4 * h + e] = sv * B_.pb[d][h][e];
        }
      int idx = 0;
#pragma unroll
      for (int bi = 0; bi < NBV; ++bi)
#pragma unroll
        for (int bj = bi; bj < NBV; ++bj) {
          acc[idx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[bi], bv[bj], acc[idx], 0, 0, 0);
          ++idx;
        }
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        const float uv = (ok && uon[j]) ? B_.pu[d][j] : 0.f;
#pragma unroll
        for (int bj = 0; bj < NBV; ++bj) acu[j][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(uv, bv[bj], acu[j][bj], 0, 0, 0);
      }
    }
  };
#ifdef BANET_TIMING
  unsigned long long tm0, tr0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm0), "=s"(tr0)::"memory");
#endif
  Batch R0, R1;
  issue(R0, q0);
  for (int q = q0; q < q1; q += 2 * kSyrkBatch) {
    issue(R1, q + kSyrkBatch);
    __builtin_amdgcn_sched_barrier(0);   // the scheduler would otherwise sink the loads next to their use
    consume(R0, q);
    __builtin_amdgcn_sched_barrier(0);
    issue(R0, q + 2 * kSyrkBatch);
    __builtin_amdgcn_sched_barrier(0);
    consume(R1, q + kSyrkBatch);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef BANET_TIMING
  unsigned long long tm1, tr1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm1), "=s"(tr1)::"memory");
#endif

  // ---- epilogue: add the 4 waves in fixed order through LDS, un-permute, publish ---------------
  for (int ww = 0; ww < kNumWaves; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int q = 0; q < NPAIR; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) sAcc[q][r][lane] = (ww == 0 ? 0.f : sAcc[q][r][lane]) + acc[q][r];
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
    if (brow < 12 && pair < PAIRS) {
#pragma unroll
      for (int bj = 0; bj < NBV; ++bj)
        part[(6 * pair + brow % 6) * K + 64 * (bj >> 2) + 4 * m + (bj & 3)] = sAcc[NPAIR + j * NBV + bj][r][lane];
    }
  }
  if (brow == 12) {   // Atb_d: rows 12 + i of block row 0 hold r_i; add the pairs in order (kq = 3, register i)
#pragma unroll
    for (int bj = 0; bj < NBV; ++bj) {
      float v = sAcc[NPAIR + bj][0][lane];
#pragma unroll
      for (int i = 1; i < PAIRS; ++i) v += sAcc[NPAIR + bj][i][lane];
      part[6 * PAIRS * K + 64 * (bj >> 2) + 4 * m + (bj & 3)] = v;
    }
  }
#ifdef BANET_TIMING
  __syncthreads();
  if (tid == 0) {   // development aid (tools/time_syrk.py): overwrites 3 words of the partial
    part[0] = (float)(tm1 - tm0);
    part[1] = (float)(tr1 - tr0);
    part[2] = (float)(q1 - q0);
  }
  return;
#endif
  float* pd = part + (6 * PAIRS + 1) * K;
  {
    int idx = 0;
    for (int bi = 0; bi < NBV; ++bi)
      for (int bj = bi; bj < NBV; ++bj) {
        const int rr = 64 * (bi >> 2) + 4 * brow + (bi & 3), cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
        const float v = sAcc[idx][r][lane];
        if (bj > bi || rr <= cc) {
          pd[rr * K + cc] = v;
          pd[cc * K + rr] = v;
        }
        ++idx;
      }
  }
}

// --------------------------------------------------------------------------------------
// ba_syrk_bf16x6_kernel -- same sums as ba_syrk_direct_kernel, H_dd on the bf16 matrix pipe at fp32
// accuracy.  s_n = jd^T M jd >= 0 (M is a Gram matrix), so H_dd = sum (sqrt(s_n) b_n)(sqrt(s_n) b_n)^T:
// A and B operands are the SAME values v = sqrt(s) b.  Each fp32 v is split EXACTLY into three bf16
// pieces (v = hi + mid + lo: 8 + 8 + 8 significand bits, v_cvt_pk_bf16_f32 and exact subtractions), and
//   v w  =  hi hi' + (hi mid' + mid hi') + (hi lo' + lo hi' + mid mid')  + O(2^-24 |v w|)
// -- six v_mfma_f32_16x16x32_bf16 (products exact in fp32, fp32 accumulate) per 32 pixels and block,
// 6 x 17 cycles against 8 x 32 cycles for v_mfma_f32_16x16x4_f32: 2.5x less matrix-pipe time, the
// dropped terms are below fp32 rounding.  Lane (m, kq) loads, for the 8 pixels 8 kq .. 8 kq + 7 of a
// 32-pixel step, the coefficients {4m..4m+3} + 64h (16-byte loads, 256 contiguous bytes per pixel and
// 16 lanes); one register quad per virtual block and piece is then directly an MFMA operand (k = pixel).
// H_cd / Atb_d stay on v_mfma_f32_16x16x4_f32 with the raw fp32 values (u is not non-negative).
// --------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// T0 = 0: six bf16 products (fp32-exact); 1: the same + this launch also leaves max_n |b_nk| per column in a.colmax (the first
// iteration of a level in the LM loop: the later ones run T0 = 16 with it, no extra pass over the basis); 3: the three largest
// bf16 products only (opt-in, see plan_syrk); 16: two fp16 pieces, three products, scaled (below)
template <int KH, int PAIRS, int T0>
__global__ __launch_bounds__(kBlock, 1) void ba_syrk_bf16x6_kernel(const SyrkArgs a) {
  constexpr int NBV = 4 * KH, NPAIR = NBV * (NBV + 1) / 2, K = 64 * KH;
  constexpr int NU = (PAIRS + 1) / 2;
  __shared__ float sAcc[NPAIR + NU * NBV][4][64];
  __shared__ float sMlp[kMlpRoleFloats];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  if (g == a.Gs) {   // the MLP role workgroup (launched only when a.mr.y != nullptr): hidden behind the SYRK workgroups
    mlp_role_block(a.mr, b, sMlp);
    return;
  }
  const int w = wave_id();
  const int N = a.N;
  const int m = lane & 15, kq = lane >> 4;
  const float* __restrict__ bas_b = a.basis + (size_t)b * N * K;
  const float* __restrict__ rec_b = a.rec + (size_t)b * PAIRS * N * 8;
  size_t uoff[NU];
  bool uon[NU];
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    int pair = -1, word = 0;
    if (m < 12) {
      pair = 2 * j + m / 6;
      word = m % 6;
    } else if (j == 0) {
      pair = m - 12;
      word = 7;
    }
    uon[j] = pair >= 0 && pair < PAIRS;
    uoff[j] = uon[j] ? (size_t)pair * N * 8 + word : 0;
  }

  f32x4 acc[NPAIR];
  f32x4 acu[NU][NBV];
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NU; ++j)
#pragma unroll
    for (int q = 0; q < NBV; ++q) acu[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- T0 = 16: the fp16 two-piece split needs every operand inside fp16's range.  v = sqrt(s_n) b_nk is scaled per COLUMN by
  // a power of two from max_n |b_nk| (ba_colmax_kernel, once per level: the basis does not change) and max_n s_n (ba_recmax_kernel,
  // per pass), so that |v| < 2^14 and a column's largest entries keep 22 significand bits while its small ones carry an absolute
  // error of 2^-25 of the scaled unit, i.e. <= 2^-39 of the column's bound; the record rows u / sqrt(s), r / sqrt(s) by one power
  // of two from max word^2 / s.  Powers of two: the scaling and its inverse (epilogue) are exact.  Inputs that are not finite, or
  // exponents beyond what the inverse can undo, fall back to the exact bf16 form below (use16 = false, wave-uniform per window).
  [[maybe_unused]] float csc[4 * KH];       // this lane's columns 64 h + 4 m + e
  [[maybe_unused]] float cmx[4 * KH];       // T0 = 1: running max |b| of those columns over this wave's pixels
  if constexpr (T0 == 1) {
#pragma unroll
    for (int c = 0; c < 4 * KH; ++c) cmx[c] = 0.f;
  }
  [[maybe_unused]] float usc = 1.f, uinv = 1.f;
  [[maybe_unused]] bool use16 = false;
  [[maybe_unused]] __shared__ float sInv[K];
  if constexpr (T0 == 16) {
    float smx = 0.f, wmx = 0.f;
    if (lane < kRecMaxBlocks) {
      smx = a.recmax[((size_t)b * kRecMaxBlocks + lane) * 2];
      wmx = a.recmax[((size_t)b * kRecMaxBlocks + lane) * 2 + 1];
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
      smx = fmaxf(smx, __shfl_xor(smx, sh, 64));
      wmx = fmaxf(wmx, __shfl_xor(wmx, sh, 64));
    }
    int es = 0, eu = 0;
    (void)frexpf(sqrtf(smx), &es);          // sqrt(s) < 2^es
    (void)frexpf(sqrtf(wmx), &eu);
    bool ok = (smx < __builtin_inff()) && (wmx < __builtin_inff()) && !(smx != smx) && !(wmx != wmx);
    usc = ldexpf(1.f, 14 - eu);
    uinv = ldexpf(1.f, eu - 14);
#pragma unroll
    for (int h = 0; h < KH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 64 * h + 4 * m + e;
        const float cm = a.colmax[(size_t)b * K + col];
        int ek = 0;
        (void)frexpf(cm, &ek);
        const int sh = 14 - ek - es;          // |sqrt(s) b| 2^sh < 2^14
        ok = ok && (cm < __builtin_inff()) && !(cm != cm) && sh > -100 && sh < 100;
        csc[4 * h + e] = ldexpf(1.f, sh);
        if (w == 0 && kq == 0) sInv[col] = ldexpf(1.f, -sh);
      }
    use16 = __ballot(!ok) == 0ull;           // the same decision in every wave of the window's workgroups (same inputs)
    __syncthreads();
  }

  // this wave's run of 32-pixel steps
  const int ns = (N + 31) >> 5, nwaves = a.Gs * kNumWaves, gw = g * kNumWaves + w;
  const int s0 = (int)(((long long)ns * gw) / nwaves), s1 = (int)(((long long)ns * (gw + 1)) / nwaves);

  f32x4 pb[8][KH];
  float ps[8][PAIRS], pu[8][NU];
  auto issue = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pix = 32 * st + 8 * kq + i;
      const size_t p = (st < s1 && pix < N) ? (size_t)pix : 0;
#pragma unroll
      for (int h = 0; h < KH; ++h)
        pb[i][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bas_b + p * K + 64 * h + 4 * m));
#pragma unroll
      for (int pr = 0; pr < PAIRS; ++pr) ps[i][pr] = rec_b[((size_t)pr * N + p) * 8 + 6];
#pragma unroll
      for (int j = 0; j < NU; ++j) pu[i][j] = rec_b[p * 8 + uoff[j]];
    }
  };
#ifdef BANET_TIMING
  unsigned long long tm0, tr0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm0), "=s"(tr0)::"memory");
#endif
  issue(s0);
  if constexpr (T0 == 16) {
    if (use16) {
      for (int st = s0; st < s1; ++st) {
        float sq[8];
        u32x4 opu[NU][2];
        {
          float ut[NU][8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool ok = 32 * st + 8 * kq + i < N;
            float ssum = ps[i][0];
#pragma unroll
            for (int pr = 1; pr < PAIRS; ++pr) ssum += ps[i][pr];
            sq[i] = ok ? sqrtf(fmaxf(ssum, 0.f)) : 0.f;          // zero switches the pixel off
            const float inv = sq[i] > 0.f ? usc / sq[i] : 0.f;
#pragma unroll
            for (int j = 0; j < NU; ++j) ut[j][i] = uon[j] ? pu[i][j] * inv : 0.f;
          }
#pragma unroll
          for (int j = 0; j < NU; ++j) split8_f16x2(ut[j], opu[j]);
        }
        u32x4 op[NBV][2];
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
        // three products, smallest first, term-major (consecutive MFMAs write different accumulators): lo hi' + hi lo' + hi hi'
        constexpr int kFa[3] = {1, 0, 0}, kFb[3] = {0, 1, 0};
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
#pragma unroll
          for (int j = 0; j < NU; ++j)
#pragma unroll
            for (int bj = 0; bj < NBV; ++bj)
              acu[j][bj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, opu[j][kFa[t3]]),
                                                                  __builtin_bit_cast(f16x8, op[bj][kFb[t3]]), acu[j][bj], 0, 0, 0);
          int idx = 0;
#pragma unroll
          for (int bi = 0; bi < NBV; ++bi)
#pragma unroll
            for (int bj = bi; bj < NBV; ++bj) {
              acc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, op[bi][kFa[t3]]),
                                                                __builtin_bit_cast(f16x8, op[bj][kFb[t3]]), acc[idx], 0, 0, 0);
              ++idx;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  for (int st = (T0 == 16 && use16) ? s1 : s0; st < s1; ++st) {
    // ---- H_cd / Atb_d on the bf16 pipe too: sum u b = sum (u / sqrt(s)) (sqrt(s) b) = sum u~ v with the SAME split v
    // as H_dd (|u| <= sqrt(Jc^T M Jc) sqrt(s): u~ is bounded, and s = 0 implies M jd = 0, i.e. u = r = 0 exactly).
    // 6 NBV bf16 MFMAs (16 cycles) per record block row instead of 8 NBV fp32 MFMAs (32 cycles).
    float sq[8];
    u32x4 opu[NU][3];
    {
      float ut[NU][8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = 32 * st + 8 * kq + i < N;
        float ssum = ps[i][0];
#pragma unroll
        for (int pr = 1; pr < PAIRS; ++pr) ssum += ps[i][pr];
        sq[i] = ok ? sqrtf(fmaxf(ssum, 0.f)) : 0.f;          // zero switches the pixel off
        const float inv = sq[i] > 0.f ? 1.f / sq[i] : 0.f;
#pragma unroll
        for (int j = 0; j < NU; ++j) ut[j][i] = uon[j] ? pu[i][j] * inv : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NU; ++j) split8_bf16x3(ut[j], opu[j]);
    }
    // ---- exact 3-way bf16 split of v = sqrt(s) b; one register quad per (virtual block, piece)
    u32x4 op[NBV][3];
#pragma unroll
    for (int h = 0; h < KH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) vv[i] = sq[i] * pb[i][h][e];
        split8_bf16x3(vv, op[4 * h + e]);
        if constexpr (T0 == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {   // (rows past N repeat row 0)  NaN-sticky like ba_colmax_kernel (fmaxf would drop a NaN and the
            const float av = fabsf(pb[i][h][e]);   // not-finite fallback of the fp16 form would then depend on who computed the maxima)
            cmx[4 * h + e] = (av > cmx[4 * h + e] || av != av) ? av : cmx[4 * h + e];
          }
        }
      }
    issue(st + 1);                                          // the raw registers are free again
    __builtin_amdgcn_sched_barrier(0);                      // keep the prefetch ahead of the MFMA block
    // Term-major order: consecutive MFMAs write DIFFERENT accumulators.  Block-major (six dependent MFMAs in a row on
    // one accumulator) measured 19 cycles per MFMA instead of 16 and kept the wave stalled at issue behind each one.
    // Per accumulator the order of the terms is unchanged (smallest first): bit-identical sums.
    constexpr int kTa[6] = {2, 0, 1, 1, 0, 0}, kTb[6] = {0, 2, 1, 0, 1, 0};
#if defined(BANET_SYRK_ABL) && BANET_SYRK_ABL >= 1   // development ablation (tools/time_syrk.py): one product instead of six
    constexpr int kT0 = 5;
#else
    constexpr int kT0 = (T0 == 16 || T0 == 1) ? 0 : T0;   // 3: mid hi' + hi mid' + hi hi' -- the third piece of the split is then dead code
#endif
#pragma unroll
    for (int t6 = kT0; t6 < 6; ++t6) {
#pragma unroll
      for (int j = 0; j < NU; ++j)
#pragma unroll
        for (int bj = 0; bj < NBV; ++bj)
          acu[j][bj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, opu[j][kTa[t6]]),
                                                               __builtin_bit_cast(bf16x8, op[bj][kTb[t6]]), acu[j][bj], 0, 0, 0);
      int idx = 0;
#pragma unroll
      for (int bi = 0; bi < NBV; ++bi)
#pragma unroll
        for (int bj = bi; bj < NBV; ++bj) {
          acc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, op[bi][kTa[t6]]),
                                                             __builtin_bit_cast(bf16x8, op[bj][kTb[t6]]), acc[idx], 0, 0, 0);
          ++idx;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef BANET_TIMING
  unsigned long long tm1, tr1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm1), "=s"(tr1)::"memory");
#endif

  if constexpr (T0 == 1) {   // column maxima: over the 4 pixel groups of the wave, then every wave of the window by atomic max
    unsigned* cm = reinterpret_cast<unsigned*>(const_cast<float*>(a.colmax)) + (size_t)b * K;     // (non-negative floats order like
#pragma unroll                                                                                   //  unsigned integers; zeroed before)
    for (int c = 0; c < 4 * KH; ++c) {
      unsigned v = __float_as_uint(cmx[c]);      // as bit patterns: |x| orders like an unsigned integer and a NaN stays the maximum
      v = max(v, (unsigned)__shfl_xor((int)v, 16, 64));
      v = max(v, (unsigned)__shfl_xor((int)v, 32, 64));
      if (kq == 0 && s1 > s0) atomicMax(&cm[64 * (c >> 2) + 4 * m + (c & 3)], v);
    }
  }
  // ---- epilogue (as ba_syrk_direct_kernel) ------------------------------------------------------
  for (int ww = 0; ww < kNumWaves; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int q = 0; q < NPAIR; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) sAcc[q][r][lane] = (ww == 0 ? 0.f : sAcc[q][r][lane]) + acc[q][r];
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
  const int r = w, brow = 4 * kq + r;
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const int pair = 2 * j + brow / 6;
    if (brow < 12 && pair < PAIRS) {
#pragma unroll
      for (int bj = 0; bj < NBV; ++bj) {
        const int cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
        float v = sAcc[NPAIR + j * NBV + bj][r][lane];
        if constexpr (T0 == 16) {
          if (use16) v = (v * uinv) * sInv[cc];          // undo the power-of-two scales: exact
        }
        part[(6 * pair + brow % 6) * K + cc] = v;
      }
    }
  }
  if (brow == 12) {
#pragma unroll
    for (int bj = 0; bj < NBV; ++bj) {
      const int cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
      float v = sAcc[NPAIR + bj][0][lane];
#pragma unroll
      for (int i = 1; i < PAIRS; ++i) v += sAcc[NPAIR + bj][i][lane];
      if constexpr (T0 == 16) {
        if (use16) v = (v * uinv) * sInv[cc];
      }
      part[6 * PAIRS * K + cc] = v;
    }
  }
#ifdef BANET_TIMING
  __syncthreads();
  if (tid == 0) {   // development aid (tools/time_syrk.py): overwrites 3 words of the partial
    part[0] = (float)(tm1 - tm0);
    part[1] = (float)(tr1 - tr0);
    part[2] = (float)(s1 - s0);
  }
  return;
#endif
  float* pd = part + (6 * PAIRS + 1) * K;
  {
    int idx = 0;
    for (int bi = 0; bi < NBV; ++bi)
      for (int bj = bi; bj < NBV; ++bj) {
        const int rr = 64 * (bi >> 2) + 4 * brow + (bi & 3), cc = 64 * (bj >> 2) + 4 * m + (bj & 3);
        float v = sAcc[idx][r][lane];
        if constexpr (T0 == 16) {
          if (use16) v = (v * sInv[rr]) * sInv[cc];
        }
        if (bj > bi || rr <= cc) {
          pd[rr * K + cc] = v;
          pd[cc * K + rr] = v;
        }
        ++idx;
      }
  }
}

// --------------------------------------------------------------------------------------
// fixed-order reduction of the per-workgroup partials of both kernels into AtA / Atb / |r| / nvalid
// (the deterministic counterpart of utils.cu:181-198 ColumnReduceSimpleKernel)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_reduce2_kernel(const float* __restrict__ gpart, int Gg, int gstride,
                                                         const float* __restrict__ spart, int Gs, int sstride,
                                                         const int32_t* active, int active_stride, int K, int C, int pairs,
                                                         float* __restrict__ AtA, float* __restrict__ Atb,
                                                         float* __restrict__ absres, float* __restrict__ nvalid) {
  // Parameter order [pose_1 .. pose_pairs | depth]; the gather partials of pair i are the rows of
  // virtual window b pairs + i.  Pose blocks of different pairs do not couple: exact zeros.
  // 64 output elements per 256-thread block; the four thread groups rq = 0..3 split the SYRK partial rows of an element into
  // four contiguous runs (a small batch has up to one row per CU: 256 dependent adds in one thread were 26 us at one window)
  // and are added ((q0 + q1) + (q2 + q3)) through LDS; the gather rows of an element are summed by group 0 alone, in the
  // order mlp.hpp's role workgroup uses.  Fixed orders: bit-reproducible.
  __shared__ float sQ[4][64];
  const int b = blockIdx.y;
  if (active != nullptr && active[(size_t)b * active_stride] == 0) return;
  const int P6 = 6 * pairs, P = P6 + K;
  const int rq = threadIdx.x >> 6, ej = threadIdx.x & 63;
  const int e = blockIdx.x * 64 + ej;
  const int total = P * P + P + C + 1;
  const bool live = e < total;
  int off = 0, pair = -1;      // pair >= 0: one pair's gather rows; -2: gather rows of all pairs; -1: syrk rows
  bool zero = false;
  float sign = 1.f;
  if (e < P * P) {
    const int i = e / P, j = e - i * P;
    if (i < P6 && j < P6) {
      if (i / 6 != j / 6) {
        zero = true;
      } else {
        pair = i / 6;
        const int ii = i % 6, jj = j % 6;
        const int lo = ii < jj ? ii : jj, hi = ii < jj ? jj : ii;
        off = 6 * lo - lo * (lo - 1) / 2 + (hi - lo);
      }
    } else if (i < P6 || j < P6) {   // H_cd: bundle sign (J = [-Jc | jd b], bundlenet.py:60)
      const int c = i < P6 ? i : j, k = (i < P6 ? j : i) - P6;
      off = c * K + k;
      sign = -1.f;
    } else {
      off = (P6 + 1) * K + (i - P6) * K + (j - P6);
    }
  } else if (e < P * P + P) {
    const int i = e - P * P;
    if (i < P6) {
      pair = i / 6;
      off = 21 + i % 6;
    } else {                         // Atb_d: d = F1 - F2w (bundlenet.py:234)
      off = P6 * K + (i - P6);
      sign = -1.f;
    }
  } else if (e < P * P + P + C) {
    pair = -2;
    off = kGHdr + (e - P * P - P);
  } else {
    pair = -2;
    off = 27;
  }
  float v = 0.f;
  const bool from_s = pair == -1;
  if (live && !zero && (from_s || rq == 0)) {
    const int p0 = pair == -2 ? 0 : pair, p1 = pair == -2 ? pairs : pair + 1;
    for (int pp = p0; pp < (from_s ? p0 + 1 : p1); ++pp) {   // fixed order: pairs, then rows
      const float* p = from_s ? spart + (size_t)b * Gs * sstride + off : gpart + ((size_t)b * pairs + pp) * Gg * gstride + off;
      const size_t st = from_s ? sstride : gstride;
      int i = 0, n = Gg;
      if (from_s) {            // this group's run of the SYRK rows
        i = (Gs * rq) >> 2;
        n = (Gs * (rq + 1)) >> 2;
      }
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (; i + 3 < n; i += 4) {
        s0 += p[(size_t)(i + 0) * st];
        s1 += p[(size_t)(i + 1) * st];
        s2 += p[(size_t)(i + 2) * st];
        s3 += p[(size_t)(i + 3) * st];
      }
      for (; i < n; ++i) s0 += p[(size_t)i * st];
      v += (s0 + s1) + (s2 + s3);
    }
    v *= sign;
  }
  sQ[rq][ej] = v;
  __syncthreads();
  if (rq != 0 || !live) return;
  if (from_s && !zero) v = (sQ[0][ej] + sQ[1][ej]) + (sQ[2][ej] + sQ[3][ej]);
  if (e < P * P)
    AtA[(size_t)b * P * P + e] = v;
  else if (e < P * P + P)
    Atb[(size_t)b * P + (e - P * P)] = v;
  else if (e < P * P + P + C)
    absres[(size_t)b * C + (e - P * P - P)] = v;
  else
    nvalid[b] = v;
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
// scales of the fp16 two-piece SYRK (ba_syrk_bf16x6_kernel<.., .., 16>)
// --------------------------------------------------------------------------------------
// per window and basis column: max_n |b_nk| as float bits (non-negative floats order like unsigned integers; a NaN ends up
// above every finite value and is caught by the consumer).  colmax zeroed before the launch.  Streams the basis once.
__global__ __launch_bounds__(256) void ba_colmax_kernel(const float* __restrict__ basis, int N, int K, const int32_t* active,
                                                        int active_stride, unsigned* __restrict__ colmax) {
  __shared__ unsigned sMax[256][4];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (active != nullptr && active[(size_t)b * active_stride] == 0) return;
  const int QK = K >> 2;                    // 16-byte column quads per row (K % 4 == 0)
  const int cq = tid % QK, r0 = tid / QK, rstep = 256 / QK;      // QK divides 256 for K = 64 / 128
  const float* bas_b = basis + (size_t)b * N * K;
  f32x4 mx = {0.f, 0.f, 0.f, 0.f};
  const int rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int n0 = blockIdx.x * rows_per_block, n1 = min(N, n0 + rows_per_block);
  for (int n = n0 + r0; n < n1; n += rstep) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(bas_b + (size_t)n * K + 4 * cq));
#pragma unroll
    for (int e = 0; e < 4; ++e) mx[e] = (fabsf(v[e]) > mx[e] || v[e] != v[e]) ? fabsf(v[e]) : mx[e];   // NaN sticks
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) sMax[tid][e] = __float_as_uint(mx[e]);
  __syncthreads();
  if (tid < QK) {
    unsigned m[4] = {0u, 0u, 0u, 0u};
    for (int r = 0; r < rstep; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = max(m[e], sMax[r * QK + tid][e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicMax(&colmax[(size_t)b * K + 4 * tid + e], m[e]);
  }
}

// per window and block: max_n s_n (s summed over the window's target frames, as the SYRK uses it) and max over pixels and
// record words (u_0..u_5 of every frame, r of every frame) of word^2 / s_n.  No atomics: [B][kRecMaxBlocks][2].
__global__ __launch_bounds__(256) void ba_recmax_kernel(const float* __restrict__ rec, int N, int pairs, const int32_t* active,
                                                        int active_stride, float* __restrict__ out) {
  __shared__ float sRed[4][2];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (active != nullptr && active[(size_t)b * active_stride] == 0) return;
  const float* rec_b = rec + (size_t)b * pairs * N * 8;
  float smx = 0.f, wmx = 0.f;
  for (int n = blockIdx.x * 256 + tid; n < N; n += gridDim.x * 256) {
    float ssum = 0.f, w2 = 0.f;
    for (int p = 0; p < pairs; ++p) {
      const float4 ua = *reinterpret_cast<const float4*>(rec_b + ((size_t)p * N + n) * 8);
      const float4 ub = *reinterpret_cast<const float4*>(rec_b + ((size_t)p * N + n) * 8 + 4);
      ssum += ub.z;
      w2 = fmaxf(w2, fmaxf(fmaxf(ua.x * ua.x, ua.y * ua.y), fmaxf(ua.z * ua.z, ua.w * ua.w)));
      w2 = fmaxf(w2, fmaxf(fmaxf(ub.x * ub.x, ub.y * ub.y), ub.w * ub.w));
      if (ua.x != ua.x || ua.y != ua.y || ua.z != ua.z || ua.w != ua.w || ub.x != ub.x || ub.y != ub.y || ub.z != ub.z || ub.w != ub.w)
        w2 = __builtin_nanf("");                                       // not finite: the consumer falls back
    }
    const float q = ssum > 0.f ? w2 / ssum : (w2 != w2 ? w2 : 0.f);   // s = 0 implies u = r = 0 exactly
    smx = (ssum > smx || ssum != ssum) ? ssum : smx;
    wmx = (q > wmx || q != q) ? q : wmx;
  }
  auto nanmax = [](float a_, float b_) { return (a_ != a_) ? a_ : (b_ != b_) ? b_ : fmaxf(a_, b_); };
#pragma unroll
  for (int sh = 1; sh < 64; sh <<= 1) {
    smx = nanmax(smx, __shfl_xor(smx, sh, 64));
    wmx = nanmax(wmx, __shfl_xor(wmx, sh, 64));
  }
  if ((tid & 63) == 0) {
    sRed[tid >> 6][0] = smx;
    sRed[tid >> 6][1] = wmx;
  }
  __syncthreads();
  if (tid == 0) {
    out[((size_t)b * kRecMaxBlocks + blockIdx.x) * 2] = nanmax(nanmax(sRed[0][0], sRed[1][0]), nanmax(sRed[2][0], sRed[3][0]));
    out[((size_t)b * kRecMaxBlocks + blockIdx.x) * 2 + 1] = nanmax(nanmax(sRed[0][1], sRed[1][1]), nanmax(sRed[2][1], sRed[3][1]));
  }
}

// --------------------------------------------------------------------------------------
static int nb_for_k(int K) {
  if (K <= 0) return 0;
  if (K <= 16) return 1;
  if (K <= 32) return 2;
  if (K <= 64) return 4;
  if (K <= 128) return 8;
  if (K <= 256) return 16;
  return -1;
}

constexpr long long kSyrkF16Pixels = 32LL * 76800;   // 320x240 x 32 windows, 640x480 x 8

int plan_syrk(int B, int Bsel, int N, int K, int pairs, int dbg, SyrkPlan* pl) {
  pl->nb = nb_for_k(K);
  if (pl->nb < 0) return BANET_ERR_UNSUPPORTED;
  if (K == 0) {
    pl->Gs = 0;
    pl->tiles = 0;
    pl->pstride = 0;
    pl->partial_bytes = 0;
    pl->f16 = pl->f16_standalone = 0;
    pl->x3 = pl->direct = 0;
    pl->off_aux = pl->off_colmax = pl->off_recmax = 0;
    return BANET_OK;
  }
  pl->tiles = (N + kTilePix - 1) / kTilePix;
  // K = 64 / 128: barrier-free kernels, one wave per SIMD, one workgroup per CU in all: 2 = ba_syrk_bf16x6_kernel
  // (default), 1 = ba_syrk_direct_kernel (fp32 MFMA; flags bit 8, A/B only)
  pl->direct = ((K == 64 || K == 128) && pairs <= 4) ? ((dbg & 256) ? 1 : 2) : 0;
  // OPT-IN (flags bit 29, K = 128): the three largest of the six products only -- a two-piece split, 16 significand bits per
  // operand, ~2^-16 relative error per product instead of 2^-24.  Not the product path and not what bench.py's `value` is measured
  // with (its dtype is "f32": fp32-exact products); measured beside it: profiles/r03_run31_*, DESIGN.md section 7.
  pl->x3 = (pl->direct == 2 && K == 128 && (dbg & (1 << 29))) ? 1 : 0;
  // K = 256, or K = 128 with more than 4 target frames: the job kernels of syrk_wide.hip (flags bit 8: the LDS-tiled kernel, A/B)
  if (!(dbg & 256) && (K == 256 || (K == 128 && pairs > 4))) pl->direct = 3;
  const int cus = num_cus();
  int target = (((pl->direct || pl->nb > 8) ? cus : 2 * cus) + Bsel - 1) / Bsel;   // Gs = partial rows per window: part of the arithmetic, hence from Bsel   // LDS kernel: 2 resident workgroups per CU (1 at K > 128)
  int G = pl->direct ? (N + 4 * 4 * 16 - 1) / (4 * 4 * 16) : pl->tiles / 4;   // direct: >= 16 quads per wave
  if (G > target) G = target;
  if (G < 1) G = 1;
  pl->Gs = G;
  pl->pstride = (int)align_up((size_t)(6 * pairs + 1) * K + (size_t)K * K, 4);
  pl->partial_bytes = align_up((size_t)B * G * pl->pstride * sizeof(float), 256);
  pl->off_aux = pl->partial_bytes;
  if (pl->direct == 3) pl->partial_bytes += syrk_wide_aux_bytes(B, N, pairs);
  // The fp16 two-piece form of ba_syrk_bf16x6_kernel (half the MFMAs, 5 instead of 9 split instructions per two values, the same
  // accuracy class: ~2^-21 per product against the reference's fp32 GEMM) where the launch is throughput-bound -- at least
  // kSyrkF16Pixels pixels in all -- so that its two small pre-passes (basis column maxima once per level, record maxima per pass)
  // are noise.  LM loop (banet_lm_level_f32) only; the single assembly pass keeps the exact bf16 form unless flags bit 24 asks
  // for this one (tests).  flags bit 31: never (A/B).
  pl->f16 = ((pl->direct == 2 || pl->direct == 3) && !pl->x3 && !(dbg & (1u << 31)) && ((long long)N * Bsel >= kSyrkF16Pixels || (dbg & (1 << 24)))) ? 1 : 0;
  pl->f16_standalone = (pl->f16 && (dbg & (1 << 24))) ? 1 : 0;
  pl->off_colmax = pl->partial_bytes;
  pl->off_recmax = pl->off_colmax + (pl->f16 ? align_up((size_t)B * K * sizeof(float), 256) : 0);
  if (pl->f16) pl->partial_bytes = pl->off_recmax + align_up((size_t)B * kRecMaxBlocks * 2 * sizeof(float), 256);
  return BANET_OK;
}

template <int NB>
static void launch_syrk_nb(const SyrkArgs& a, int B, hipStream_t s) {
  constexpr int KPAD = NB * 16, KP = KPAD + 16;
  size_t fl = (size_t)2 * kTilePix * KP + 2 * kTilePix * kUStrideS;
  const size_t epi = (size_t)4 * 8 * KPAD;
  if (fl < epi) fl = epi;
  const size_t lds = fl * sizeof(float);
  auto k = ba_syrk_kernel<NB>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(a.Gs, B), dim3(kBlock), lds, s, a);
}

template <int KH, int T0>
static void launch_bf16x6(const SyrkArgs& a, int B, hipStream_t s) {
  const dim3 grid(a.Gs + (a.mr.y != nullptr ? 1 : 0), B), block(kBlock);
  switch (a.pairs) {
    case 1: hipLaunchKernelGGL((ba_syrk_bf16x6_kernel<KH, 1, T0>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((ba_syrk_bf16x6_kernel<KH, 2, T0>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((ba_syrk_bf16x6_kernel<KH, 3, T0>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((ba_syrk_bf16x6_kernel<KH, 4, T0>), grid, block, 0, s, a); break;
  }
}

template <int KH>
static void launch_direct(const SyrkArgs& a, int B, hipStream_t s) {
  const dim3 grid(a.Gs, B), block(kBlock);
  switch (a.pairs) {
    case 1: hipLaunchKernelGGL((ba_syrk_direct_kernel<KH, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((ba_syrk_direct_kernel<KH, 2>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((ba_syrk_direct_kernel<KH, 3>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((ba_syrk_direct_kernel<KH, 4>), grid, block, 0, s, a); break;
  }
}

int launch_syrk(const float* basis, const float* rec, int B, int N, int K, int pairs, const SyrkPlan& pl,
                const int32_t* active, int active_stride, float* partials, hipStream_t s, const MlpRole* mr, int f16_stats) {
  if (pl.direct == 3) {
    const float* colmax_w = nullptr;
    float* recmax_w = nullptr;
    // The wide jobs differ from ba_syrk_bf16x6_kernel in ONE respect of the f16_stats contract (kernels.hpp): they have no variant that
    // harvests the column maxima in-kernel, so f16_stats = 2 ("first iteration of a level") runs ba_colmax_kernel as a pass of its own
    // and the fp16 form already on that iteration; 0 does the same for a one-iteration call, 1 = maxima in place.
    if (pl.f16 && f16_stats >= 0) {      // the job kernels' fp16 two-piece form: column maxima by their own pass at the level's first call
      unsigned* cm = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(partials) + pl.off_colmax);
      recmax_w = reinterpret_cast<float*>(reinterpret_cast<char*>(partials) + pl.off_recmax);
      if (f16_stats != 1) {
        launch_zero_iters(reinterpret_cast<int32_t*>(cm), B * K, s);
        const int G = std::max(1, std::min((N + 255) / 256, (8 * num_cus() + B - 1) / B));
        hipLaunchKernelGGL(ba_colmax_kernel, dim3(G, B), dim3(256), 0, s, basis, N, K, active, active_stride, cm);
      }
      hipLaunchKernelGGL(ba_recmax_kernel, dim3(kRecMaxBlocks, B), dim3(256), 0, s, rec, N, pairs, active, active_stride, recmax_w);
      colmax_w = reinterpret_cast<const float*>(cm);
    }
    return launch_syrk_wide(basis, rec, B, N, K, pairs, pl.Gs, pl.pstride, active, active_stride, partials,
                            reinterpret_cast<float*>(reinterpret_cast<char*>(partials) + pl.off_aux), s, colmax_w, recmax_w);
  }
  SyrkArgs a{basis, rec, active, active_stride, partials, N, K, pl.Gs, pl.tiles, pl.pstride, pairs, 0, MlpRole{}, nullptr, nullptr};
  if (pl.direct == 2) {
    if (mr != nullptr) a.mr = *mr;
    if (pl.f16 && f16_stats == 2) {     // the exact form, which also leaves the basis column maxima for the passes that follow
      unsigned* colmax = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(partials) + pl.off_colmax);
      launch_zero_iters(reinterpret_cast<int32_t*>(colmax), B * K, s);
      a.colmax = reinterpret_cast<const float*>(colmax);
      if (K == 128)
        launch_bf16x6<2, 1>(a, B, s);
      else
        launch_bf16x6<1, 1>(a, B, s);
      return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
    }
    if (pl.f16 && f16_stats >= 0) {     // fp16 two-piece form: f16_stats 0 = compute the basis column maxima now, 1 = they are in place
      unsigned* colmax = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(partials) + pl.off_colmax);
      float* recmax = reinterpret_cast<float*>(reinterpret_cast<char*>(partials) + pl.off_recmax);
      if (f16_stats == 0) {
        launch_zero_iters(reinterpret_cast<int32_t*>(colmax), B * K, s);
        const int G = std::max(1, std::min((N + 255) / 256, (8 * num_cus() + B - 1) / B));
        hipLaunchKernelGGL(ba_colmax_kernel, dim3(G, B), dim3(256), 0, s, basis, N, K, active, active_stride, colmax);
      }
      hipLaunchKernelGGL(ba_recmax_kernel, dim3(kRecMaxBlocks, B), dim3(256), 0, s, rec, N, pairs, active, active_stride, recmax);
      a.colmax = reinterpret_cast<const float*>(colmax);
      a.recmax = recmax;
      if (K == 128)
        launch_bf16x6<2, 16>(a, B, s);
      else
        launch_bf16x6<1, 16>(a, B, s);
      return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
    }
    if (K == 128) {
      if (pl.x3)
        launch_bf16x6<2, 3>(a, B, s);
      else
        launch_bf16x6<2, 0>(a, B, s);
    } else {
      launch_bf16x6<1, 0>(a, B, s);
    }
    return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
  }
  if (pl.direct) {
    if (K == 128)
      launch_direct<2>(a, B, s);
    else
      launch_direct<1>(a, B, s);
    return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
  }
  for (int pass = 0; pass < pairs; ++pass) {   // LDS-tiled kernel: one pass per pair (H_dd / Atb_d in pass 0)
    a.pass = pass;
    switch (pl.nb) {
      case 1: launch_syrk_nb<1>(a, B, s); break;
      case 2: launch_syrk_nb<2>(a, B, s); break;
      case 4: launch_syrk_nb<4>(a, B, s); break;
      case 8: launch_syrk_nb<8>(a, B, s); break;
      case 16: launch_syrk_nb<16>(a, B, s); break;
      default: return BANET_ERR_UNSUPPORTED;
    }
  }
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

void launch_reduce2(const float* gpart, int Gg, int gstride, const float* spart, int Gs, int sstride,
                    const int32_t* active, int active_stride, int B, int K, int C, int pairs, float* AtA, float* Atb,
                    float* absres, float* nvalid, hipStream_t s) {
  const int P = 6 * pairs + K, total = P * P + P + C + 1;
  hipLaunchKernelGGL(ba_reduce2_kernel, dim3((total + 63) / 64, B), dim3(256), 0, s, gpart, Gg, gstride, spart, Gs,
                     sstride, active, active_stride, K, C, pairs, AtA, Atb, absres, nvalid);
}

}  // namespace banet
