// ba_gather128_kernel -- the gather pass specialised for the reference's feature width C = 128
// (legacy/feat.py:251, dec.py:174) with the gradient computed on the fly from the C-channel
// target map.  Same arithmetic as ba_gather_kernel (gather.hip); different lane mapping and
// scheduling:
//
//   * 16 lanes per pixel, 8 channels per lane: a wave instruction works on FOUR pixels (a 2x2
//     block of the 8x8 patch, so their 12-texel stencils overlap and hit in L1), every texel row
//     is fetched as 2 x 16-byte loads per lane (the widest, cheapest VMEM form), and all
//     per-pixel overhead (addresses, weights, reductions) is amortised over 4 pixels;
//   * the five channel sums per pixel are reduced inside one 16-lane DPP row
//     (row_ror:8, row_half_mirror, quad_perm x2): no LDS crossbar, no cross-row traffic;
//   * per-pixel parameters go from the geometry phase (lane = pixel) to the gather phase
//     (lane = pixel-group x channel-slice) through a 2 KB per-wave LDS table: wave-private, so
//     there is no workgroup barrier anywhere -- a workgroup is just 4 independent waves;
//   * the kernel is latency-bound per wave (measured: a wave's batch takes the same time with 1
//     or 3 workgroups on the CU), so the grid is ONE resident round (plan_gather) and every wave
//     pulls 8x8 tiles from a per-(window, XCD band) atomic queue until all bands are empty: no
//     tail round, no static imbalance;
//   * every tile publishes its own partial (28 pose sums via the transposing butterfly + C x
//     sum|d|); ba_fold_kernel / ba_reduce2_kernel add the tile partials in tile order, so the
//     result does not depend on which wave happened to process which tile (bit-reproducible).
#include "gather_common.hpp"

namespace banet {

constexpr int kC128 = 128;
constexpr int kParStride = 8;  // src offset, texel offset, w00,w01,w10,w11 (pre-masked), mask, -

__device__ __forceinline__ void morton8(int n, int& px, int& py) {  // pixel id -> position in the 8x8 patch
  px = (n & 1) | ((n >> 1) & 2) | ((n >> 2) & 4);
  py = ((n >> 1) & 1) | ((n >> 2) & 2) | ((n >> 3) & 4);
}

__device__ __forceinline__ int brev5(int t) { return (int)(__brev((unsigned)t) >> 27); }

// carry chain of the transposing butterfly over NL levels, first lane distance S0 (cf. carry_push)
template <int NL, int S0>
__device__ __forceinline__ void carry_push_n(float (&pend)[NL + 1], float v, int t) {
  bool done = false;
#pragma unroll
  for (int L = 0; L < NL; ++L) {
    if (!done) {
      if (((t >> L) & 1) == 0) {
        pend[L] = v;
        done = true;
      } else {
        v = bfly_merge(pend[L], v, S0 >> L);
      }
    }
  }
  if (!done) pend[NL] = v;
}

// KV4 = number of 128-coefficient chunks of a basis row (0: pose only; K % 4 == 0, K <= 128 KV4)
template <int KV4>
__global__ __launch_bounds__(kBlock, BANET_G128_WAVES) void ba_gather128_kernel(const GatherArgs a) {
  __shared__ __attribute__((aligned(16))) float sPar[kNumWaves][64][kParStride];
  __shared__ __attribute__((aligned(16))) float sQ[kNumWaves][64][8];
  __shared__ float sAbs[kNumWaves][kC128];
  const banet_level_t& lv = a.lv;
  const int vb = blockIdx.y, g = blockIdx.x;   // vb = (window, pair): a multi-frame window is `pairs` virtual windows
  const int b = vb / a.pairs;                  // that share the key frame's source map, depth, basis and Wc
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id();
  const int N = lv.N, K = lv.K, H = lv.H, W = lv.W;
  constexpr int C = kC128;
  const bool dense = lv.dense != 0;
  const float* __restrict__ tgt_b = lv.tgt + (size_t)vb * H * W * C;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ dep_b = lv.depth + (size_t)b * N;
  const float* __restrict__ bas_b = KV4 ? lv.basis + (size_t)b * N * K : nullptr;
  float* __restrict__ rec_b = KV4 ? a.rec + (size_t)vb * N * 8 : nullptr;
  const int qshift = a.qshift;                       // 2: a work item is one quarter (16 pixels, 4 steps) of a tile
  const int nitems = a.tiles << qshift;
  float* __restrict__ part_b = a.partials + (size_t)vb * nitems * (kGHdr + C);
  const int grp = lane >> 4, sub = lane & 15;
  const int half = lane >> 5, li = lane & 31;

  float wreg[KV4 ? KV4 : 1][4];  // this lane's slice of the depth coefficients
  if constexpr (KV4 > 0) {
#pragma unroll
    for (int kc = 0; kc < KV4; ++kc)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kc * 128 + li * 4 + e;
        wreg[kc][e] = (k < K) ? a.Wc[(size_t)b * K + k] : 0.f;
      }
  }

  // ---- tile queue: band x = tiles [tiles x / nb, tiles (x+1) / nb); home band = this workgroup's XCD
  const int nb = a.nbands;
  int* __restrict__ queue = a.queue + vb * 8;
  int band = nb > 1 ? (g & 7) : 0, left = nb;
  auto band_lo = [&](int x) { return (int)(((long long)nitems * x) / nb); };
  auto pop_raw = [&](int x) {  // issues the atomic; the id is read (v_readfirstlane) only when it is needed
    int v = 0;
    if (lane == 0) v = atomicAdd(&queue[x], 1);
    return v;
  };
  int raw_next = pop_raw(band);

  while (true) {
    int wi = rfl(raw_next) + band_lo(band);
    while (wi >= band_lo(band + 1)) {  // this band is drained: move on (a drained band stays drained)
      if (--left == 0) return;
      band = band + 1 == nb ? 0 : band + 1;
      wi = rfl(pop_raw(band)) + band_lo(band);
    }
    // Small levels (fewer tiles than resident waves) are latency-bound: there a tile is split into 4
    // work items that each redo the tile's (cheap) depth dot and geometry but gather only their own
    // 16 pixels -- 4x the waves, a quarter of the serial step chain.
    const int t = wi >> qshift;
    const int s_lo = qshift ? 4 * (wi & 3) : 0, s_hi = qshift ? s_lo + 4 : 16;
    const bool mine = (lane >> 2) >= s_lo && (lane >> 2) < s_hi;   // lane = pixel: pixels 4 s .. 4 s + 3 belong to step s
    raw_next = pop_raw(band);  // issued now, read at the top of the next tile: the atomic's latency is hidden
    BANET_TICK(tb0);
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
    float absd8[8];   // |d| of channels {4 sub + e, 64 + 4 sub + e} over this lane group's pixels
#pragma unroll
    for (int i = 0; i < 8; ++i) absd8[i] = 0.f;
    float absd2[1][2] = {{0.f, 0.f}};  // rim pixels (generic routine: channels 2 lane, 2 lane + 1)

    // ---- 1. depth: D_j = D0_j + b_j . W.  A half wave reads one basis row per instruction
    // (16 B per lane); 32 row pairs go through a 5-level transposing butterfly inside each half,
    // leaf t of half h carrying pixel h*32 + brev5(t), so that pixel j's sum lands on lane j.
    float D = valid ? dep_b[pt] : 0.f;
    if constexpr (KV4 > 0) {
      float pend[6];
#pragma unroll
      for (int q16 = 0; q16 < 2; ++q16) {
        float part[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          bool vj;
          const int ptj = point_of(half * 32 + brev5(q16 * 16 + i), vj);
          const float* row = bas_b + (size_t)ptj * K;
          float acc = 0.f;
#pragma unroll
          for (int kc = 0; kc < KV4; ++kc) {
            const int k = kc * 128 + li * 4;
            const bool ok = k < K;
            const f32x4 bv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + (ok ? k : 0)));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(bv[e], wreg[kc][e], acc);   // k >= K: the address is clamped to the row's own first quad and wreg is 0
          }
          part[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) carry_push_n<5, 16>(pend, part[i], q16 * 16 + i);
      }
      D += pend[5];
    }
    BANET_TICK(tb1);

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
      const float* Rm = a.R + vb * 9;
      const float* Tv = a.T + vb * 3;
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

    BANET_TICK(tb2);
    // ---- 3. gather: 16 steps x 4 pixels; lane = (pixel group, 8-channel slice) --------------
    const int rowC = W * C;
    for (int s = s_lo; s < s_hi; ++s) {
      const int j = 4 * s + grp;
      const float4 pa = *reinterpret_cast<const float4*>(&sPar[w][j][0]);
      const float4 pb = *reinterpret_cast<const float4*>(&sPar[w][j][4]);
#ifdef BANET_ABLATE   // development aid (tools/prof_assemble.py): flags bit 0 -> every tap reads texel (1,1) / point 0
      const bool abl = (lv.flags & 1) != 0;
      const unsigned osrc = abl ? 0u : (unsigned)__float_as_int(pa.x), oa = abl ? (unsigned)((W + 1) * C) : (unsigned)__float_as_int(pa.y);
#else
      const unsigned osrc = (unsigned)__float_as_int(pa.x), oa = (unsigned)__float_as_int(pa.y);
#endif
      const float w00 = pa.z, w01 = pa.w, w10 = pb.x, w11 = pb.y, mk = pb.z;
      const float h00 = 0.5f * w00, h01 = 0.5f * w01, h10 = 0.5f * w10, h11 = 0.5f * w11;
      typedef float v2f_ __attribute__((ext_vector_type(2)));
      v2f_ qm11 = {0.f, 0.f}, qm12 = {0.f, 0.f}, qm22 = {0.f, 0.f}, qg1 = {0.f, 0.f}, qg2 = {0.f, 0.f};
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
        // packed fp32 (v_pk_fma_f32 / v_pk_add_f32): two channels per VALU instruction.  The 0.5 of
        // the central difference is folded into the weights (exact: a power of two) and the mask
        // into the residual's fma (mk is 0 or 1; f is already masked through the weights).
        typedef float v2f __attribute__((ext_vector_type(2)));
#define BANET_V2(v, k) (v2f){(k) ? (v).z : (v).x, (k) ? (v).w : (v).y}
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const v2f F1 = BANET_V2(f1, k);
          const v2f A0 = BANET_V2(a0, k), A1 = BANET_V2(a1, k), A2 = BANET_V2(a2, k), A3 = BANET_V2(a3, k);
          const v2f B0 = BANET_V2(b0, k), B1 = BANET_V2(b1, k), B2 = BANET_V2(b2, k), B3 = BANET_V2(b3, k);
          const v2f M1 = BANET_V2(m1, k), M2 = BANET_V2(m2, k), P1 = BANET_V2(p1, k), P2 = BANET_V2(p2, k);
          const v2f f = ((A1 * w00 + A2 * w01) + B1 * w10) + B2 * w11;
          const v2f gx = (((A2 - A0) * h00 + (A3 - A1) * h01) + (B2 - B0) * h10) + (B3 - B1) * h11;
          const v2f gy = (((B1 - M1) * h00 + (B2 - M2) * h01) + (P1 - A1) * h10) + (P2 - A2) * h11;
          const v2f d = f - F1 * mk;
          qm11 += gx * gx;
          qm12 += gx * gy;
          qm22 += gy * gy;
          qg1 += gx * d;
          qg2 += gy * d;
          absd8[h * 4 + 2 * k] += fabsf(d.x);
          absd8[h * 4 + 2 * k + 1] += fabsf(d.y);
        }
#undef BANET_V2
      }
      Q5 q;
      q.m11 = qm11.x + qm11.y;
      q.m12 = qm12.x + qm12.y;
      q.m22 = qm22.x + qm22.y;
      q.g1 = qg1.x + qg1.y;
      q.g2 = qg2.x + qg2.y;
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
      unsigned long long slow = __ballot((gflags & 4) != 0 && mine);
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

    BANET_TICK(tb3);

    // ---- 4. per-pixel 6x6 algebra (lane = pixel), then the tile's 28 pose sums --------------
    float* __restrict__ part = part_b + (size_t)wi * (kGHdr + C);
    if (!mine) {   // pixels of the other quarters: exact zeros (sQ holds stale sums for them)
      q = Q5{0.f, 0.f, 0.f, 0.f, 0.f};
      gflags = 0;
    }
    {
      float mj[12];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        mj[i] = q.m11 * jc[i] + q.m12 * jc[6 + i];
        mj[6 + i] = q.m12 * jc[i] + q.m22 * jc[6 + i];
      }
      // leaves 0..20: upper triangle of Jc^T M Jc, 21..26: Jc^T g, 27: valid count, 28..31: zero.
      // 5 levels (lane distance 32..2) + one xor-1 add: lane l ends with leaf brev5(l >> 1).
      float pend[6];
      int o = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jj = i; jj < 6; ++jj) {
          carry_push_n<5, 32>(pend, jc[i] * mj[jj] + jc[6 + i] * mj[6 + jj], o);
          ++o;
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) carry_push_n<5, 32>(pend, jc[i] * q.g1 + jc[6 + i] * q.g2, 21 + i);
      carry_push_n<5, 32>(pend, (float)(gflags & 1), 27);
      if (a.mask_out != nullptr && valid && mine) a.mask_out[(size_t)vb * N + pt] = (unsigned char)(gflags & 1);
#pragma unroll
      for (int i = 28; i < 32; ++i) carry_push_n<5, 32>(pend, 0.f, i);
      float tot = pend[5];
      tot += dpp_mov<kDppXor1>(tot);
      const int leaf = brev5(lane >> 1);
      if ((lane & 1) == 0 && leaf < 28) part[leaf] = tot;

      if constexpr (KV4 > 0) {
        if (valid && mine) {
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

    // ---- 5. the tile's C x sum|d| ------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // fold the 4 pixel groups (fixed order), group 0 publishes
      float v = absd8[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (grp == 0) sAbs[w][(i >> 2) * 64 + 4 * sub + (i & 3)] = v;
    }
    sAbs[w][2 * lane] += absd2[0][0];       // same wave: LDS operations retire in program order
    sAbs[w][2 * lane + 1] += absd2[0][1];
    part[kGHdr + lane] = sAbs[w][lane];
    part[kGHdr + 64 + lane] = sAbs[w][64 + lane];
#ifdef BANET_TIMING
    {
      BANET_TICK(tb9);
      if (lane == 0) {
        part[28] = (float)(tb3 - tb2);  // 16 gather steps
        part[29] = (float)(tb9 - tb3);  // rim patch + algebra + partial + rec store
        part[30] = (float)(tb1 - tb0);  // depth dot
        part[31] = (float)(tb9 - tb0);  // whole tile
      }
    }
#endif
  }  // tiles
}

int launch_gather128(const GatherArgs& a, int K, hipStream_t s) {
  dim3 grid(a.G, a.lv.B * a.pairs), block(kBlock);
  if (K == 0)
    hipLaunchKernelGGL((ba_gather128_kernel<0>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 128)
    hipLaunchKernelGGL((ba_gather128_kernel<1>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 256)
    hipLaunchKernelGGL((ba_gather128_kernel<2>), grid, block, 0, s, a);
  else
    return BANET_ERR_UNSUPPORTED;
  return BANET_OK;
}

}  // namespace banet
