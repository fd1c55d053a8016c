// ba_gather128_kernel -- the gather pass specialised for the reference's feature width C = 128
// (legacy/feat.py:251, dec.py:174) with the gradient computed on the fly from the C-channel
// target map.  Same arithmetic as ba_gather_kernel (gather.hip); different lane mapping:
//
//   * 16 lanes per pixel, 8 channels per lane: a wave instruction works on FOUR pixels (a 2x2
//     block of the 8x8 patch, so their 12-texel stencils overlap and hit in L1), every texel row
//     is fetched as 2 x 16-byte loads per lane (the widest, cheapest VMEM form), and all
//     per-pixel overhead (addresses, weights, reductions) is amortised over 4 pixels;
//   * the five channel sums per pixel are reduced inside one 16-lane DPP row
//     (row_ror:8, row_half_mirror, quad_perm x2): no LDS crossbar, no cross-row traffic;
//   * per-pixel parameters go from the geometry phase (lane = pixel) to the gather phase
//     (lane = pixel-group x channel-slice) through a 2 KB per-wave LDS table: wave-private,
//     so still no workgroup barrier anywhere in the main loop.
#include "gather_common.hpp"

namespace banet {

constexpr int kC128 = 128;
constexpr int kParStride = 8;  // src offset, texel offset, w00,w01,w10,w11 (pre-masked), mask, -

__device__ __forceinline__ void morton8(int n, int& px, int& py) {  // pixel id -> position in the 8x8 patch
  px = (n & 1) | ((n >> 1) & 2) | ((n >> 2) & 4);
  py = ((n >> 1) & 1) | ((n >> 2) & 2) | ((n >> 3) & 4);
}

template <int KVEC, int KCH>
__global__ __launch_bounds__(kBlock, 3) void ba_gather128_kernel(const GatherArgs a) {
  __shared__ float sH[kNumWaves][28][64];                 // per-lane H_cc / Atb_c / nvalid accumulators
  __shared__ float sAbs[kNumWaves][kC128];
  __shared__ __attribute__((aligned(16))) float sPar[kNumWaves][64][kParStride];
  __shared__ __attribute__((aligned(16))) float sQ[kNumWaves][64][8];
  const banet_level_t& lv = a.lv;
  const int b = blockIdx.y, g = blockIdx.x;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id();
  const int N = lv.N, K = lv.K, H = lv.H, W = lv.W;
  constexpr int C = kC128;
  const bool dense = lv.dense != 0;
  const float* __restrict__ tgt_b = lv.tgt + (size_t)b * H * W * C;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ dep_b = lv.depth + (size_t)b * N;
  const float* __restrict__ bas_b = KCH ? lv.basis + (size_t)b * N * K : nullptr;
  float* __restrict__ rec_b = KCH ? a.rec + (size_t)b * N * 8 : nullptr;
  const int grp = lane >> 4, sub = lane & 15;

#pragma unroll
  for (int i = 0; i < 28; ++i) sH[w][i][lane] = 0.f;
  float absd8[8];   // |d| of channels {4 sub + e, 64 + 4 sub + e} over this lane group's pixels
#pragma unroll
  for (int i = 0; i < 8; ++i) absd8[i] = 0.f;
  float absd2[1][2] = {{0.f, 0.f}};  // rim pixels (generic routine: channels 2 lane, 2 lane + 1)

  float wreg[KCH ? KCH : 1][KVEC];
  if constexpr (KCH > 0) {
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int e = 0; e < KVEC; ++e) {
        const int k = (kc * 64 + lane) * KVEC + e;
        wreg[kc][e] = (k < K) ? a.Wc[(size_t)b * K + k] : 0.f;
      }
  }

  int s_begin, s_end, s_step;
  if ((a.G & 7) == 0) {
    const int x = g & 7, s = g >> 3, per = a.G >> 3;
    s_begin = (int)(((long long)a.groups * x) >> 3) + s;
    s_end = (int)(((long long)a.groups * (x + 1)) >> 3);
    s_step = per;
  } else {
    s_begin = g;
    s_end = a.groups;
    s_step = a.G;
  }

  for (int sg = s_begin; sg < s_end; sg += s_step) {
    const int t = sg * 4 + w;
    if (t >= a.tiles) continue;  // wave-uniform
    int tx = 0, ty = 0;
    if (dense) tile_coords(t, a.tiles_x, a.tiles_y, tx, ty);
    auto point_of = [&](int n, bool& valid) -> int {
      if (dense) {
        int qx, qy;
        morton8(n, qx, qy);
        const int py = (ty << 3) + qy, px = (tx << 3) + qx;
        valid = (py < H) && (px < W);
        return valid ? py * W + px : 0;
      }
      const int pt = t * kTilePix + n;
      valid = pt < N;
      return valid ? pt : 0;
    };
    bool valid;
    const int pt = point_of(lane, valid);

    // ---- 1. depth: D_j = D0_j + b_j . W  (64 coalesced row loads, transposing butterfly) ----
    float D = valid ? dep_b[pt] : 0.f;
    if constexpr (KCH > 0) {
      float pend[6], dsum = 0.f;
      for (int q8 = 0; q8 < 8; ++q8) {
        float part[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          bool vj;
          const int ptj = point_of(brev6(q8 * 8 + i), vj);   // leaf t ends on lane brev6(t)
          const float* row = bas_b + (size_t)ptj * K;
          float acc = 0.f;
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            const int k = (kc * 64 + lane) * KVEC;
            const Vec<KVEC> bv = ldv_nt<KVEC>(row, k, k < K);
#pragma unroll
            for (int e = 0; e < KVEC; ++e) acc = fmaf(bv.v[e], wreg[kc][e], acc);
          }
          part[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) carry_push(pend, part[i], q8 * 8 + i, dsum);
      }
      D += dsum;
    }

    // ---- 2. geometry, lane = pixel -------------------------------------------------------
    float gw00 = 0.f, gw01 = 0.f, gw10 = 0.f, gw11 = 0.f, jd0 = 0.f, jd1 = 0.f;
    float jc[12];
    int gx0 = 1, gy0 = 1, gflags = 0;
    {
      float p0 = 0.f, p1 = 0.f, p2 = 1.f, fx = 1.f, fy = 1.f, ox = 0.f, oy = 0.f;
      if (valid) {
        if (dense) {
          const float fx0 = lv.intr[b * 4 + 0], fy0 = lv.intr[b * 4 + 1], ox0 = lv.intr[b * 4 + 2],
                      oy0 = lv.intr[b * 4 + 3];
          const int py = pt / W, px = pt - py * W;
          p0 = ((float)px * lv.scale - ox0) / fx0;
          p1 = ((float)py * lv.scale - oy0) / fy0;
          p2 = 1.f;
          if (lv.normalize_rays) {
            const float ss = p0 * p0 + p1 * p1 + p2 * p2;
            const float inv = 1.f / sqrtf(fmaxf(ss, 1e-12f));
            p0 *= inv;
            p1 *= inv;
            p2 *= inv;
          }
          fx = fx0 / lv.scale;
          fy = fy0 / lv.scale;
          ox = ox0 / lv.scale;
          oy = oy0 / lv.scale;
        } else {
          const size_t o = (size_t)b * 3 * N;
          p0 = lv.rays[o + pt];
          p1 = lv.rays[o + N + pt];
          p2 = lv.rays[o + 2 * (size_t)N + pt];
          const size_t q = (size_t)b * N + pt;
          fx = lv.fx[q];
          fy = lv.fy[q];
          ox = lv.ox[q];
          oy = lv.oy[q];
        }
      }
      const float* Rm = a.R + b * 9;
      const float* Tv = a.T + b * 3;
      const float rx = Rm[0] * p0 + Rm[1] * p1 + Rm[2] * p2;
      const float ry = Rm[3] * p0 + Rm[4] * p1 + Rm[5] * p2;
      const float rz = Rm[6] * p0 + Rm[7] * p1 + Rm[8] * p2;
      const float X = rx * D + Tv[0], Y = ry * D + Tv[1], Z = rz * D + Tv[2];
      const float x = X / Z, y = Y / Z;
      const float pxl = fx * x + ox, pyl = fy * y + oy;
      const bool m = valid && (pxl >= 0.f) && (pxl <= (float)(W - 1)) && (pyl >= 0.f) && (pyl <= (float)(H - 1));
#pragma unroll
      for (int i = 0; i < 12; ++i) jc[i] = 0.f;
      int x0 = 0, y0 = 0;
      if (m) {
        const float xf = floorf(pxl), yf = floorf(pyl);
        const float dx = pxl - xf, dy = pyl - yf;
        x0 = (int)xf;
        y0 = (int)yf;
        gw00 = (1.f - dx) * (1.f - dy);
        gw01 = dx * (1.f - dy);
        gw10 = (1.f - dx) * dy;
        gw11 = dx * dy;
        const float iz = 1.f / Z;
        jc[0] = fx * (x * y);
        jc[1] = fx * (-1.f - x * x);
        jc[2] = fx * y;
        jc[3] = fx * (-iz);
        jc[4] = 0.f;
        jc[5] = fx * (x / Z);
        jc[6] = fy * (1.f + y * y);
        jc[7] = fy * (-(x * y));
        jc[8] = fy * (-x);
        jc[9] = 0.f;
        jc[10] = fy * (-iz);
        jc[11] = fy * (y / Z);
        jd0 = fx * ((rx - rz * x) / Z);
        jd1 = fy * ((ry - rz * y) / Z);
      }
      const bool interior = (x0 >= 1) && (x0 + 2 <= W - 1) && (y0 >= 1) && (y0 + 2 <= H - 1);
      const bool fast = m && interior;
      gflags = (m ? 1 : 0) | (fast ? 2 : 0) | ((m && !fast) ? 4 : 0);
      gx0 = x0;
      gy0 = y0;
      // parameters of the branch-free gather: non-fast pixels read the safe texel (1,1) with
      // zero weights
      const float mk = fast ? 1.f : 0.f;
      float4 pa, pb;
      pa.x = __int_as_float(pt * C);
      pa.y = __int_as_float(((fast ? y0 : 1) * W + (fast ? x0 : 1)) * C);
      pa.z = mk * gw00;
      pa.w = mk * gw01;
      pb.x = mk * gw10;
      pb.y = mk * gw11;
      pb.z = mk;
      pb.w = 0.f;
      *reinterpret_cast<float4*>(&sPar[w][lane][0]) = pa;
      *reinterpret_cast<float4*>(&sPar[w][lane][4]) = pb;
    }

    // ---- 3. gather: 16 steps x 4 pixels; lane = (pixel group, 8-channel slice) --------------
    const int rowC = W * C;
    for (int s = 0; s < 16; ++s) {
      const int j = 4 * s + grp;
      const float4 pa = *reinterpret_cast<const float4*>(&sPar[w][j][0]);
      const float4 pb = *reinterpret_cast<const float4*>(&sPar[w][j][4]);
      const unsigned osrc = (unsigned)__float_as_int(pa.x), oa = (unsigned)__float_as_int(pa.y);
      const float w00 = pa.z, w01 = pa.w, w10 = pb.x, w11 = pb.y, mk = pb.z;
      Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned co = 64u * h + 4u * sub;
        const float* ra = tgt_b + (size_t)(oa + co);   // texel (y0, x0), this lane's 4 channels
        const float* rb = ra + rowC;                   // row y0 + 1
        const float* rm = ra - rowC;                   // row y0 - 1
        const float* rp = rb + rowC;                   // row y0 + 2
        const f32x4 f1v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src_b + (size_t)(osrc + co)));  // streamed once: keep it out of L2's way
        const float4 f1 = make_float4(f1v[0], f1v[1], f1v[2], f1v[3]);
        const float4 a0 = *reinterpret_cast<const float4*>(ra - C), a1 = *reinterpret_cast<const float4*>(ra),
                     a2 = *reinterpret_cast<const float4*>(ra + C), a3 = *reinterpret_cast<const float4*>(ra + 2 * C);
        const float4 b0 = *reinterpret_cast<const float4*>(rb - C), b1 = *reinterpret_cast<const float4*>(rb),
                     b2 = *reinterpret_cast<const float4*>(rb + C), b3 = *reinterpret_cast<const float4*>(rb + 2 * C);
        const float4 m1 = *reinterpret_cast<const float4*>(rm), m2 = *reinterpret_cast<const float4*>(rm + C);
        const float4 p1 = *reinterpret_cast<const float4*>(rp), p2 = *reinterpret_cast<const float4*>(rp + C);
        const float F1[4] = {f1.x, f1.y, f1.z, f1.w};
        const float A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w};
        const float A2[4] = {a2.x, a2.y, a2.z, a2.w}, A3[4] = {a3.x, a3.y, a3.z, a3.w};
        const float B0[4] = {b0.x, b0.y, b0.z, b0.w}, B1[4] = {b1.x, b1.y, b1.z, b1.w};
        const float B2[4] = {b2.x, b2.y, b2.z, b2.w}, B3[4] = {b3.x, b3.y, b3.z, b3.w};
        const float M1[4] = {m1.x, m1.y, m1.z, m1.w}, M2[4] = {m2.x, m2.y, m2.z, m2.w};
        const float P1[4] = {p1.x, p1.y, p1.z, p1.w}, P2[4] = {p2.x, p2.y, p2.z, p2.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f = ((A1[e] * w00 + A2[e] * w01) + B1[e] * w10) + B2[e] * w11;
          const float gx00 = 0.5f * (A2[e] - A0[e]), gx01 = 0.5f * (A3[e] - A1[e]);
          const float gx10 = 0.5f * (B2[e] - B0[e]), gx11 = 0.5f * (B3[e] - B1[e]);
          const float gx = ((gx00 * w00 + gx01 * w01) + gx10 * w10) + gx11 * w11;
          const float gy00 = 0.5f * (B1[e] - M1[e]), gy01 = 0.5f * (B2[e] - M2[e]);
          const float gy10 = 0.5f * (P1[e] - A1[e]), gy11 = 0.5f * (P2[e] - A2[e]);
          const float gy = ((gy00 * w00 + gy01 * w01) + gy10 * w10) + gy11 * w11;
          const float d = mk * (f - F1[e]);
          q.m11 = fmaf(gx, gx, q.m11);
          q.m12 = fmaf(gx, gy, q.m12);
          q.m22 = fmaf(gy, gy, q.m22);
          q.g1 = fmaf(gx, d, q.g1);
          q.g2 = fmaf(gy, d, q.g2);
          absd8[h * 4 + e] += fabsf(d);
        }
      }
      q.m11 = row16_sum(q.m11);
      q.m12 = row16_sum(q.m12);
      q.m22 = row16_sum(q.m22);
      q.g1 = row16_sum(q.g1);
      q.g2 = row16_sum(q.g2);
      if (sub == 0) {
        *reinterpret_cast<float4*>(&sQ[w][j][0]) = make_float4(q.m11, q.m12, q.m22, q.g1);
        sQ[w][j][4] = q.g2;
      }
    }
    Q5 q;
    {
      const float4 qa = *reinterpret_cast<const float4*>(&sQ[w][lane][0]);
      q.m11 = qa.x;
      q.m12 = qa.y;
      q.m22 = qa.z;
      q.g1 = qa.w;
      q.g2 = sQ[w][lane][4];
    }
    {
      // patch the pixels whose stencil touches the image rim (rare): generic slow routine
      unsigned long long slow = __ballot((gflags & 4) != 0);
      while (slow) {  // wave-uniform
        const int j = __builtin_ctzll(slow);
        slow &= slow - 1;
        Q5 e = border_pixel_q5<2, 1>(rdl(gx0, j), rdl(gy0, j), rdl(gw00, j), rdl(gw01, j), rdl(gw10, j), rdl(gw11, j),
                                     src_b + (size_t)rdl(pt, j) * C, tgt_b, C, H, W, lane, absd2);
        e.m11 = wave_sum(e.m11);
        e.m12 = wave_sum(e.m12);
        e.m22 = wave_sum(e.m22);
        e.g1 = wave_sum(e.g1);
        e.g2 = wave_sum(e.g2);
        if (lane == j) {
          q.m11 += e.m11;
          q.m12 += e.m12;
          q.m22 += e.m22;
          q.g1 += e.g1;
          q.g2 += e.g2;
        }
      }
    }

    // ---- 4. per-pixel 6x6 algebra, lane = pixel ------------------------------------------
    {
      float mj[12];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        mj[i] = q.m11 * jc[i] + q.m12 * jc[6 + i];
        mj[6 + i] = q.m12 * jc[i] + q.m22 * jc[6 + i];
      }
      int o = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jj = i; jj < 6; ++jj) {
          atomicAdd(&sH[w][o][lane], jc[i] * mj[jj] + jc[6 + i] * mj[6 + jj]);
          ++o;
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) atomicAdd(&sH[w][21 + i][lane], jc[i] * q.g1 + jc[6 + i] * q.g2);
      atomicAdd(&sH[w][27][lane], (float)(gflags & 1));
      if constexpr (KCH > 0) {
        if (valid) {
          const float md0 = q.m11 * jd0 + q.m12 * jd1, md1 = q.m12 * jd0 + q.m22 * jd1;
          float4 ua, ub;
          ua.x = jc[0] * md0 + jc[6] * md1;
          ua.y = jc[1] * md0 + jc[7] * md1;
          ua.z = jc[2] * md0 + jc[8] * md1;
          ua.w = jc[3] * md0 + jc[9] * md1;
          ub.x = jc[4] * md0 + jc[10] * md1;
          ub.y = jc[5] * md0 + jc[11] * md1;
          ub.z = jd0 * md0 + jd1 * md1;    // s_n
          ub.w = jd0 * q.g1 + jd1 * q.g2;  // r_n
          float4* rp = reinterpret_cast<float4*>(rec_b + (size_t)pt * 8);
          rp[0] = ua;
          rp[1] = ub;
        }
      }
    }
  }  // tiles

  // ---- epilogue: one small partial per workgroup -------------------------------------------
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // fold the 4 pixel groups (fixed order), group 0 publishes
    float v = absd8[i];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (grp == 0) sAbs[w][(i >> 2) * 64 + 4 * sub + (i & 3)] = v;
  }
  sAbs[w][2 * lane] += absd2[0][0];       // same wave: LDS operations retire in program order
  sAbs[w][2 * lane + 1] += absd2[0][1];
  __syncthreads();
  float* __restrict__ part = a.partials + ((size_t)b * a.G + g) * (kGHdr + C);
  if (tid < 4 * 28) {
    const int ww = tid / 28, i = tid - ww * 28;
    float s = 0.f;
    for (int l = 0; l < 64; ++l) s += sH[ww][i][l];
    sH[ww][i][0] = s;
  }
  __syncthreads();
  if (tid < 28) part[tid] = (sH[0][tid][0] + sH[1][tid][0]) + (sH[2][tid][0] + sH[3][tid][0]);
  for (int c = tid; c < C; c += kBlock) part[kGHdr + c] = (sAbs[0][c] + sAbs[1][c]) + (sAbs[2][c] + sAbs[3][c]);
}

int launch_gather128(const GatherArgs& a, int K, hipStream_t s) {
  dim3 grid(a.G, a.lv.B), block(kBlock);
  const bool keven = (K & 1) == 0;
  if (K == 0)
    hipLaunchKernelGGL((ba_gather128_kernel<1, 0>), grid, block, 0, s, a);
  else if (keven && K <= 128)
    hipLaunchKernelGGL((ba_gather128_kernel<2, 1>), grid, block, 0, s, a);
  else if (keven && K <= 256)
    hipLaunchKernelGGL((ba_gather128_kernel<2, 2>), grid, block, 0, s, a);
  else if (K <= 64)
    hipLaunchKernelGGL((ba_gather128_kernel<1, 1>), grid, block, 0, s, a);
  else if (K <= 128)
    hipLaunchKernelGGL((ba_gather128_kernel<1, 2>), grid, block, 0, s, a);
  else
    return BANET_ERR_UNSUPPORTED;
  return BANET_OK;
}

}  // namespace banet
