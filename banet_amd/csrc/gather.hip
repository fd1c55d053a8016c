// ba_gather_kernel -- the HBM-bound half of one BA assembly pass on gfx950.
//
// Per pixel n of window b (citations under /root/reference):
//   D = D0 + Bs.W                         bundlenet.py:208
//   X = (R p) D + T ; x,y,Z ; px,py       bundlenet.py:209-224 / legacy/ba.py:239-251
//   sample [f|gx|gy] at (px,py), mask     utils_python.py:61-117 / bundlenet.py:154-157
//   d = m (F2w - F1) ; G = m [gx,gy]      legacy/ba.py:258-264 (bundlenet: opposite sign)
//   M = G^T G (2x2), g = G^T d            utils.cu:331-340,382-391 (GEMM-F1, F4)
//   Jc (2x6), jd (2)                      legacy/ba.py:36-48 ; bundlenet.py:49-74
//   H_cc += Jc^T M Jc ; Atb_c += Jc^T g   utils.cu:344-380,393-414 restricted to the pose block
//   record (u = Jc^T M jd, s = jd^T M jd, r = jd^T g) for the depth-basis blocks (syrk.hip)
//
// This file: the GENERIC kernel (any C <= 256, K <= 256, sparse points, the reference's precomputed [f|gx|gy]
// target layout) plus the planning / dispatch of all gather kernels; the C = 128 dense fast paths live in
// gather128.hip (direct loads) and gather128p.hip (wave-private LDS patches, large levels).
// Design: NO workgroup barriers and no LDS in the main loop -- every wave is an independent
// stream, so occupancy (3 waves/SIMD) hides the dependent chain basis-row -> depth -> projection
// -> texel address -> 13 row loads.  A wave owns 64-pixel batches (8x8 patches):
//   1. depth: the 64 basis rows are read as coalesced 512-B rows (64 loads in flight), each lane
//      keeps its partial dot with W, and a 6-level transposing butterfly (one shuffle per value)
//      leaves D_j on lane j;
//   2. geometry with lane = pixel (all 64 lanes busy);
//   3. gather with lane = channel pair: pixel j's scalars come from v_readlane (SGPRs), its
//      texel rows are coalesced 512-B loads with scalar bases; two pixels per trip = 26 loads in
//      flight per wave; the five channel sums per pixel go through the same butterfly so that
//      pixel j's sums land on lane j;
//   4. 6x6 algebra with lane = pixel; H_cc in LDS accumulators (one owner per address, plain read-modify-write).
#include <algorithm>
#include "gather_common.hpp"

#ifndef BANET_GATHER_WAVES
#define BANET_GATHER_WAVES 3
#endif

namespace banet {

// VEC/CH: channels per lane / channel chunks (C <= 64*VEC*CH); GRAD: target is the 3C [f|gx|gy]
// map; KVEC/KCH: basis coefficients per lane / chunks (K <= 64*KVEC*KCH), KCH = 0: no basis.
template <int VEC, int CH, bool GRAD, int KVEC, int KCH>
__global__ __launch_bounds__(kBlock, BANET_GATHER_WAVES) void ba_gather_kernel(const GatherArgs a) {
  __shared__ float sH[kNumWaves][28][64];   // per-wave, per-lane accumulators of H_cc / Atb_c / nvalid
  __shared__ float sAbs[kNumWaves][256];
  const banet_level_t& lv = a.lv;
  const int vb = blockIdx.y, g = blockIdx.x;   // vb = (window, pair), see gather128.hip
  const int b = vb / a.pairs;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id();
  const int N = lv.N, C = lv.C, K = lv.K, H = lv.H, W = lv.W;
  const int Ct = GRAD ? 3 * C : C;
  const bool dense = lv.dense != 0;
  const int dbg = lv.flags;  // profiling ablation bits (tools/prof_assemble.py); 0 in production
  const float* __restrict__ tgt_b = lv.tgt + (size_t)vb * H * W * Ct;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ dep_b = lv.depth + (size_t)b * N;
  const float* __restrict__ bas_b = KCH ? lv.basis + (size_t)b * N * K : nullptr;
  float* __restrict__ rec_b = KCH ? a.rec + (size_t)vb * N * 8 : nullptr;

#pragma unroll
  for (int i = 0; i < 28; ++i) sH[w][i][lane] = 0.f;
#ifdef BANET_TIMING
  float tim0 = 0.f, tim1 = 0.f, tim2 = 0.f, tim3 = 0.f;
#endif
  float absd[CH][VEC];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) absd[i][e] = 0.f;

  // this lane's share of the coefficient vector W
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

  // schedule: groups of 4 consecutive tiles (one per wave); groups are dealt to workgroups in 8
  // XCD bands when the grid allows it, so that a band's tiles share L2
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
    auto point_of = [&](int n, bool& valid) -> int {  // n uniform or per lane
      if (dense) {
        int qx, qy;
        patch_pos(n, qx, qy);
        const int py = (ty << 3) + qy, px = (tx << 3) + qx;
        valid = (py < H) && (px < W);
        return valid ? py * W + px : 0;
      }
      const int pt = t * a.tile_pts + n;        // sparse points: a.tile_pts (64, or 16 on latency-bound launches) per wave item
      valid = n < a.tile_pts && pt < N;
      return valid ? pt : 0;
    };
    // slots of this item that hold a point (wave-uniform); the others contribute exact zeros and are skipped
    const int npts = dense ? 64 : min(a.tile_pts, N - t * a.tile_pts);
    bool valid;
    const int pt = point_of(lane, valid);
    BANET_TICK(tb0);

    // ---- 1. depth: D_j = D0_j + b_j . W ---------------------------------------------
    float D = valid ? dep_b[pt] : 0.f;
    if constexpr (KCH > 0) {
      float pend[6], dsum = 0.f;
      for (int q8 = 0; q8 < ((dbg & 4) ? 0 : 8); ++q8) {
        float part[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          bool vj;
          const int ptj = point_of(brev6(q8 * 8 + i), vj);
          const float* row = bas_b + (size_t)ptj * K;
          float acc = 0.f;
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            const int k = (kc * 64 + lane) * KVEC;
            const Vec<KVEC> bv = ldv<KVEC>(row, k, k < K);
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

#ifdef BANET_TIMING
    asm volatile("" ::"v"(D));
#endif
    BANET_TICK(tb1);
    BANET_TACC(tim2, tb0, tb1);
    // ---- 2. geometry, lane = pixel ----------------------------------------------------
    float gw00 = 0.f, gw01 = 0.f, gw10 = 0.f, gw11 = 0.f, jd0 = 0.f, jd1 = 0.f;
    float jc[12];
    int gx0 = 1, gy0 = 1, gflags = 0, sx0 = 1, sy0 = 1;
    float smk = 0.f;
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
      // fast path: 3C map (taps clamped one by one) or interior stencil; rim pixels of the
      // on-the-fly gradient go to the slow path and look masked to the main loop
      const bool interior = (x0 >= 1) && (x0 + 2 <= W - 1) && (y0 >= 1) && (y0 + 2 <= H - 1);
      const bool fast = m && (GRAD || interior);
      gflags = (m ? 1 : 0) | (fast ? 2 : 0) | ((m && !fast) ? 4 : 0);
      gx0 = x0;
      gy0 = y0;
      sx0 = fast ? x0 : 1;
      sy0 = fast ? y0 : 1;
      smk = fast ? 1.f : 0.f;
    }

    // ---- 3. gather: pixel j's scalars from lane j, rows as coalesced 512-B loads -----------
    auto pixel_q5 = [&](int j) __attribute__((always_inline)) -> Q5 {
      Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
      const int x0 = (dbg & 1) ? 1 : rdl(sx0, j), y0 = (dbg & 1) ? 1 : rdl(sy0, j);  // safe interior texel when not on the fast path
      const float mk = rdl(smk, j);
      const float w00 = mk * rdl(gw00, j), w01 = mk * rdl(gw01, j), w10 = mk * rdl(gw10, j), w11 = mk * rdl(gw11, j);
      const int ptj = rdl(pt, j);
      const float* __restrict__ srow = src_b + (size_t)ptj * C;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const int c = (ch * 64 + lane) * VEC;
        const bool ok = c < C;
        const Vec<VEC> f1 = ldv<VEC>(srow, c, ok);
        Vec<VEC> f, gx, gy;
        if constexpr (GRAD) {
          const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x0 + 1, 0), W - 1);
          const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y0 + 1, 0), H - 1);
          const float* r00 = tgt_b + (size_t)(y0c * W + x0c) * Ct;
          const float* r01 = tgt_b + (size_t)(y0c * W + x1c) * Ct;
          const float* r10 = tgt_b + (size_t)(y1c * W + x0c) * Ct;
          const float* r11 = tgt_b + (size_t)(y1c * W + x1c) * Ct;
          const Vec<VEC> a00 = ldv<VEC>(r00, c, ok), a01 = ldv<VEC>(r01, c, ok), a10 = ldv<VEC>(r10, c, ok), a11 = ldv<VEC>(r11, c, ok);
          const Vec<VEC> b00 = ldv<VEC>(r00 + C, c, ok), b01 = ldv<VEC>(r01 + C, c, ok), b10 = ldv<VEC>(r10 + C, c, ok), b11 = ldv<VEC>(r11 + C, c, ok);
          const Vec<VEC> c00 = ldv<VEC>(r00 + 2 * C, c, ok), c01 = ldv<VEC>(r01 + 2 * C, c, ok), c10 = ldv<VEC>(r10 + 2 * C, c, ok), c11 = ldv<VEC>(r11 + 2 * C, c, ok);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            f.v[e] = ((a00.v[e] * w00 + a01.v[e] * w01) + a10.v[e] * w10) + a11.v[e] * w11;
            gx.v[e] = ((b00.v[e] * w00 + b01.v[e] * w01) + b10.v[e] * w10) + b11.v[e] * w11;
            gy.v[e] = ((c00.v[e] * w00 + c01.v[e] * w01) + c10.v[e] * w10) + c11.v[e] * w11;
          }
        } else {
          // the 4 taps and their +-1 neighbours are 12 distinct texels (interior stencil)
          const float* ra = tgt_b + (size_t)(y0 * W + x0) * C;      // row y0, col x0
          const float* rb = ra + (size_t)W * C;                     // row y0+1
          const float* rm = ra - (size_t)W * C;                     // row y0-1
          const float* rp = rb + (size_t)W * C;                     // row y0+2
          const Vec<VEC> a0 = ldv<VEC>(ra - C, c, ok), a1 = ldv<VEC>(ra, c, ok), a2 = ldv<VEC>(ra + C, c, ok), a3 = ldv<VEC>(ra + 2 * C, c, ok);
          const Vec<VEC> b0 = ldv<VEC>(rb - C, c, ok), b1 = ldv<VEC>(rb, c, ok), b2 = ldv<VEC>(rb + C, c, ok), b3 = ldv<VEC>(rb + 2 * C, c, ok);
          const Vec<VEC> m1 = ldv<VEC>(rm, c, ok), m2 = ldv<VEC>(rm + C, c, ok);
          const Vec<VEC> p1 = ldv<VEC>(rp, c, ok), p2 = ldv<VEC>(rp + C, c, ok);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            f.v[e] = ((a1.v[e] * w00 + a2.v[e] * w01) + b1.v[e] * w10) + b2.v[e] * w11;
            const float gx00 = 0.5f * (a2.v[e] - a0.v[e]), gx01 = 0.5f * (a3.v[e] - a1.v[e]);
            const float gx10 = 0.5f * (b2.v[e] - b0.v[e]), gx11 = 0.5f * (b3.v[e] - b1.v[e]);
            gx.v[e] = ((gx00 * w00 + gx01 * w01) + gx10 * w10) + gx11 * w11;
            const float gy00 = 0.5f * (b1.v[e] - m1.v[e]), gy01 = 0.5f * (b2.v[e] - m2.v[e]);
            const float gy10 = 0.5f * (p1.v[e] - a1.v[e]), gy11 = 0.5f * (p2.v[e] - a2.v[e]);
            gy.v[e] = ((gy00 * w00 + gy01 * w01) + gy10 * w10) + gy11 * w11;
          }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = mk * (f.v[e] - f1.v[e]);
          q.m11 = fmaf(gx.v[e], gx.v[e], q.m11);
          q.m12 = fmaf(gx.v[e], gy.v[e], q.m12);
          q.m22 = fmaf(gy.v[e], gy.v[e], q.m22);
          q.g1 = fmaf(gx.v[e], d, q.g1);
          q.g2 = fmaf(gy.v[e], d, q.g2);
          absd[ch][e] += fabsf(d);
        }
      }
      return q;
    };

    Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
    {
      Q5 pend[6];
      for (int tt = 0; tt < ((dbg & 8) ? 0 : 64); tt += 2) {
        const int j0 = brev6(tt);          // leaf tt+1 is pixel j0 + 32
        BANET_TICK(t0);
        const Q5 zero5{0.f, 0.f, 0.f, 0.f, 0.f};
        const Q5 qa = j0 < npts ? pixel_q5(j0) : zero5;
        const Q5 qb = j0 + 32 < npts ? pixel_q5(j0 + 32) : zero5;
#ifdef BANET_TIMING
        asm volatile("" ::"v"(qa.m11), "v"(qb.m11), "v"(qa.g2), "v"(qb.g2));
#endif
        BANET_TICK(t1);
        BANET_TACC(tim0, t0, t1);
        Q5 m1 = q5_merge(qa, qb, 32);      // level 0
        // levels 1..5 of the carry chain on leaf-pair index tt/2
        const int tp = tt >> 1;
        bool done = false;
#pragma unroll
        for (int L = 1; L < 6; ++L) {
          if (!done) {
            if (((tp >> (L - 1)) & 1) == 0) {
              pend[L] = m1;
              done = true;
            } else {
              m1 = q5_merge(pend[L], m1, 32 >> L);
            }
          }
        }
        if (!done) q = m1;
#ifdef BANET_TIMING
        asm volatile("" ::"v"(m1.m11), "v"(m1.g2));
#endif
        BANET_TICK(t2);
        BANET_TACC(tim1, t1, t2);
      }
    }
    if constexpr (!GRAD) {
      // patch the pixels whose stencil touches the image rim (rare)
      unsigned long long slow = __ballot((gflags & 4) != 0);
      while (slow) {  // wave-uniform
        const int j = __builtin_ctzll(slow);
        slow &= slow - 1;
        Q5 e = border_pixel_q5<VEC, CH>(rdl(gx0, j), rdl(gy0, j), rdl(gw00, j), rdl(gw01, j), rdl(gw10, j),
                                        rdl(gw11, j), src_b + (size_t)rdl(pt, j) * C, tgt_b, C, H, W, lane, absd);
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

    // ---- 4. per-pixel 6x6 algebra, lane = pixel ---------------------------------------
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
        for (int j = i; j < 6; ++j) {
          sH[w][o][lane] += jc[i] * mj[j] + jc[6 + i] * mj[6 + j];      // one owner per address: plain read-modify-write
          ++o;                                                           // (ds_add_f32 measured ~240 ns each, round 6)
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) sH[w][21 + i][lane] += jc[i] * q.g1 + jc[6 + i] * q.g2;
      sH[w][27][lane] += (float)(gflags & 1);
      if (a.mask_out != nullptr && valid) a.mask_out[(size_t)vb * N + pt] = (unsigned char)(gflags & 1);
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
    BANET_TICK(tb9);
    BANET_TACC(tim3, tb0, tb9);
  }  // tiles
#ifdef BANET_TIMING
  if (tid == 0) {
    float* dp = a.partials + ((size_t)vb * a.G + g) * (kGHdr + C);
    dp[28] = tim0;  // cycles: two pixels' loads + channel math
    dp[29] = tim1;  // cycles: butterfly merges
    dp[30] = tim2;  // cycles: depth dot
    dp[31] = tim3;  // cycles: whole batch
  }
#endif

  // ---- epilogue: one small partial per workgroup -------------------------------------------
#pragma unroll
  for (int ch = 0; ch < CH; ++ch)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = (ch * 64 + lane) * VEC + e;
      if (c < C) sAbs[w][c] = absd[ch][e];
    }
  __syncthreads();
  float* __restrict__ part = a.partials + ((size_t)vb * a.G + g) * (kGHdr + C);
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

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
constexpr int kStrip8PerWave = 2;     // 8-row strip segments: fewest per resident wave (tools/gpu_r4_k.sh)
constexpr int kQuadRounds = 12;      // ba_gather128q_kernel: most items per resident wave (measured, profiles/r04_run6_*: wins up to 19200 items -- 640x480 x 1, 160x120 x 8, 80x60 x 32 -- loses at 38400)
constexpr int kStripSegW = 16, kStripSegH = 32, kStripMinW = 21;   // = kStripW, kStripH, kWinTex of strip_plan.hpp (gather128s.hip)
constexpr int kGenericBlocksPerCU = BANET_GATHER_WAVES;  // ba_gather_kernel: launch bounds
constexpr int kC128BlocksPerCU = BANET_G128_WAVES;       // ba_gather128_kernel: launch bounds (LDS: 18 KB)

static bool use_c128(const banet_level_t* lv) {
  // flags bit 5 (A/B experiments only): force the generic kernel
  return lv->C == 128 && !lv->tgt_has_grad && !(lv->flags & 32) && (lv->K & 3) == 0 && lv->K <= 256;
}

int plan_gather(const banet_level_t* lv, GatherPlan* pl) {
  const int kCUs = num_cus();
  const int Bsel = lv ? selection_batch(lv) : 0;   // decisions: Bsel; grids and buffer sizes: lv->B
  if (!lv || lv->B <= 0 || lv->N <= 0 || lv->C <= 0 || lv->K < 0 || lv->H < 4 || lv->W < 4) return BANET_ERR_INVALID_ARG;
  // (the field was a must-be-zero pad until round 4: anything but the two policies is a caller's garbage, not "throughput")
  if (lv->policy != BANET_POLICY_THROUGHPUT && lv->policy != BANET_POLICY_BATCH_INVARIANT) return BANET_ERR_INVALID_ARG;
  if (lv->C > 256 || lv->K > 256) return BANET_ERR_UNSUPPORTED;
  if (lv->dense && lv->N != lv->H * lv->W) return BANET_ERR_INVALID_ARG;
  if (lv->dense) {
    pl->tiles_x = (lv->W + 7) / 8;
    pl->tiles_y = (lv->H + 7) / 8;
    pl->tiles = pl->tiles_x * pl->tiles_y;
  } else {
    pl->tiles_x = pl->tiles_y = 0;
    // Sparse points: 64 per wave item -- or 16 (round 6) while four times as many items still fit one resident round: the item is
    // a serial chain of its points (the reference's own tracker, N = 4096 at batch 1: 64 items = 64 waves on 16 CUs, 54 us per launch)
    const long long items64 = (long long)((lv->N + kTilePix - 1) / kTilePix) * Bsel * npairs(lv);
    pl->tile_pts = (4 * items64 <= (long long)kCUs * kGenericBlocksPerCU * kNumWaves && !(lv->flags & 2)) ? 16 : kTilePix;   // bit 1: 64 (A/B)
    pl->tiles = (lv->N + pl->tile_pts - 1) / pl->tile_pts;
  }
  if (lv->dense) pl->tile_pts = kTilePix;
  pl->groups = (pl->tiles + 3) / 4;
  pl->c128 = use_c128(lv) ? 1 : 0;
  // The strip gather (gather128s.hip: 16 x 32-pixel segments, rolling LDS window, target map fetched 1.44 x instead of 2.15 x)
  // where a launch has at least 4 segments per resident wave (coarser items than the 8x8 tiles: below that the tail of the
  // last round costs more than the halo saves).  flags bit 18: force it at any size (parity tests); bit 19: off (A/B).
  pl->strip = 0;
  pl->strip_fp = 0;
  if (pl->c128 && lv->dense && !(lv->flags & 524288) && lv->W >= kStripMinW && lv->W < 4096 && lv->H < 4096 &&
      (size_t)lv->N * lv->C * 4 < ((size_t)1 << 31)) {
    // segment height: 16 rows.  32-row segments fetch less (target rows 35/32 x instead of 19/16 x: launch 1.12 x vs 1.16 x the
    // algorithmic bytes) but lose 5 % at every size measured (640x480 x 32: 3396 vs 3230 us, x 256: 27.3 vs 26.1 ms, 5-frame
    // windows 11.03 vs 10.77 ms; profiles/r03_run11_*, r03_run12_*): half as many, twice as long work items leave a longer
    // tail in the last round of the queue.  flags bit 21: 32-row segments (A/B, parity tests).
    const int sxn = (lv->W + kStripSegW - 1) / kStripSegW;
    // Mid-size two-frame launches (too few 16-row segments per wave, too many pixels for the 4x4-item kernel: 160x120 x 32,
    // 320x240 x 8 .. 16, 640x480 x 2 .. 4): 8-row segments (target rows 11/8 x) where the launch has at least kStrip8PerWave of
    // them per resident wave.  flags bits 18 + 10: force them (parity tests); A/B: bit 19 (no strip gather at all).
    const int np = npairs(lv);
    const int syn16 = (lv->H + kStripSegH / 2 - 1) / (kStripSegH / 2), syn8 = (lv->H + 7) / 8;
    const bool force8 = (lv->flags & 262144) && (lv->flags & 1024);
    const bool low = force8 || (np == 1 && !(lv->flags & ((1 << 21) | 262144)) &&
                                (long long)sxn * syn16 * Bsel < 4LL * kCUs * 8 &&
                                (long long)sxn * syn8 * Bsel >= (long long)kStrip8PerWave * kCUs * 8 &&
                                (long long)((lv->W + 3) / 4) * ((lv->H + 3) / 4) * Bsel > (long long)kQuadRounds * kCUs * 8);
    const int segh = (lv->flags & (1 << 21)) ? kStripSegH : low ? 8 : kStripSegH / 2;
    const int syn = (lv->H + segh - 1) / segh;
    // (A single resident round at tiny batches -- every segment on a wave of its own -- does not pay: one segment is a serial chain
    // of ~200-270 us whatever the load; batch 1: 640x480 270 vs 272 us, 320x240 205 vs 168 us for the tile kernels.)
    // multi-frame windows: frame-parallel workgroups (gather128s.hip, FP) -- `pairs` waves per segment, so a launch has
    // 2048 / pairs resident work-item slots instead of 2048.  flags bit 22: the frames looped over inside one wave (A/B).
    const bool fp = np >= 2 && np <= 7 && segh == kStripSegH / 2 && !(lv->flags & (1 << 22));
    const int fp_wg_per_cu = fp ? (int)std::min<size_t>(8 / np, (size_t)(160 * 1024) / (((size_t)np * (7 * 21 * 32 + 128) + 4 * 64 + 4) * 4)) : 0;
    const long long slots = fp ? (long long)kCUs * fp_wg_per_cu : (long long)kCUs * 8;
    pl->strip_fp = 0;
    if ((long long)sxn * syn * Bsel >= 4LL * slots || low || (lv->flags & 262144)) {
      pl->strip = segh;
      pl->strip_fp = fp ? 1 : 0;
      pl->quad = 0;
      pl->tiles_x = sxn;
      pl->tiles_y = syn;
      pl->tiles = sxn * syn;
      pl->patch = 0;
      pl->pairloop = 1;
      pl->qshift = 0;
      const int resident = fp ? kCUs * fp_wg_per_cu : kCUs * 4;   // 128-thread workgroups, 4 per CU (2 waves per SIMD); FP: by LDS / waves
      int G = (resident + lv->B - 1) / lv->B;
      const int want = fp ? pl->tiles : (pl->tiles + 1) / 2;      // one item per wave (FP: per workgroup) at least
      if (G > want) G = want;
      if (G < 1) G = 1;
      pl->G = G;
      pl->nbands = 1;
      pl->pstride = kGHdr + lv->C;
      pl->rows = pl->tiles;
      pl->frows = pl->rows > kFoldRows ? (pl->rows + kFoldRows - 1) / kFoldRows : pl->rows;
      const int VBs = lv->B * npairs(lv);
      const size_t row_bytes_s = (size_t)VBs * pl->pstride * sizeof(float);
      pl->off_fold = align_up(row_bytes_s * pl->rows, 256);
      pl->off_queue = pl->off_fold + (pl->frows != pl->rows ? align_up(row_bytes_s * pl->frows, 256) : 0);
      pl->partial_bytes = pl->off_queue + align_up((size_t)VBs * 8 * sizeof(int), 256);
      pl->rec_bytes = lv->K > 0 ? align_up((size_t)VBs * lv->N * 8 * sizeof(float), 256) : 0;
      return BANET_OK;
    }
  }
  // Latency-bound launches (coarse levels, small batches): ba_gather128q_kernel -- 4x4-pixel items, the whole item one step,
  // ~4x shorter serial chain per item than a tile's 16 steps (gather128q.hip) -- while the launch has at most kQuadRounds items
  // per resident wave (2 workgroups x 4 waves per CU); beyond that the tile kernels' shared stencils win.
  // flags bit 25: force it at any size (parity tests, A/B); bit 30: off.
  pl->quad = 0;
  if (pl->c128 && lv->dense && !(lv->flags & ((1 << 30) | 512 | 64))) {   // (bits 9 / 6 force the patch / direct tile kernels)
    const int qxn = (lv->W + 3) / 4, qyn = (lv->H + 3) / 4;
    const long long qitems = (long long)qxn * qyn * Bsel * npairs(lv);
    // (multi-frame windows: every virtual window redoes the depth dot -- half the limit: cfg-3's 40x30 x 32 x 4 = 9600 items wins,
    //  cfg-5's 80x60 x 8 x 7 = 16800 loses 329 vs 210 us, profiles/r04_run9_*)
    const long long qlimit = (long long)kQuadRounds * kCUs * 8 / (npairs(lv) > 1 ? 2 : 1);
    if (qitems <= qlimit || (lv->flags & (1 << 25))) {
      const int VBq = lv->B * npairs(lv);
      pl->quad = 1;
      pl->patch = 0;
      pl->pairloop = 0;
      pl->qshift = 0;
      pl->tiles_x = qxn;
      pl->tiles_y = qyn;
      pl->tiles = qxn * qyn;
      int G = (kCUs * 2 + VBq - 1) / VBq;            // one resident round of 256-thread workgroups, 2 per CU
      const int want = (pl->tiles + kNumWaves - 1) / kNumWaves;
      if (G > want) G = want;
      if (G < 1) G = 1;
      pl->G = G;
      pl->nbands = 1;
      pl->pstride = kGHdr + lv->C;
      pl->rows = pl->tiles;
      pl->frows = pl->rows > kFoldRows ? (pl->rows + kFoldRows - 1) / kFoldRows : pl->rows;
      const size_t row_bytes_q = (size_t)VBq * pl->pstride * sizeof(float);
      pl->off_fold = align_up(row_bytes_q * pl->rows, 256);
      pl->off_queue = pl->off_fold + (pl->frows != pl->rows ? align_up(row_bytes_q * pl->frows, 256) : 0);
      pl->partial_bytes = pl->off_queue + align_up((size_t)VBq * 8 * sizeof(int), 256);
      pl->rec_bytes = lv->K > 0 ? align_up((size_t)VBq * lv->N * 8 * sizeof(float), 256) : 0;
      return BANET_OK;
    }
  }
  // One resident round: the gather is latency-bound per wave (measured: a second, partial round of
  // workgroups takes as long as the first), so the grid is what the chip holds at once, split
  // across the windows.
  // Large dense levels (>= 4 tiles per resident wave): ba_gather128p_kernel -- wave-private LDS patches with a
  // software-pipelined box prefetch, 2 workgroups per CU (measured 640x480 x 8: 138 -> 125 us/window, 320x240 x 8:
  // 48 -> 38); smaller levels are latency-bound and keep the 3-per-CU direct kernel.  flags bit 6: direct (A/B).
  pl->patch = (pl->c128 && lv->dense && !(lv->flags & 64) &&
               ((long long)pl->tiles * Bsel * npairs(lv) >= 4LL * kCUs * BANET_G128P_WAVES * kNumWaves ||
                (lv->flags & 512))) ? 1 : 0;   // bit 9: force it at any size (parity tests)
  const int resident = kCUs * (pl->patch ? BANET_G128P_WAVES : pl->c128 ? kC128BlocksPerCU : kGenericBlocksPerCU);
  const int VB = lv->B * npairs(lv);   // virtual windows
  const int VBsel = Bsel * npairs(lv);
  // the generic kernel publishes one partial row per WORKGROUP (rows = G), so its grid is part of the arithmetic: taken from Bsel;
  // the C = 128 kernels publish one row per work item in a fixed place, their grid is free to follow the launch
  const int Bgrid = pl->c128 ? lv->B : Bsel;
  // the patch kernel loops over a window's target frames inside a tile (depth dot once per window) where a window alone has
  // >= 4 tiles per resident wave; below that the 4x coarser items cost more than the shared depth saves (80x60 x 32
  // windows x 4 frames: 327 -> 478 us) and a work item stays one pair's tile
  pl->pairloop = (pl->patch && npairs(lv) > 1 && ((long long)pl->tiles * Bsel >= 4LL * kCUs * BANET_G128P_WAVES * kNumWaves ||
                                                   (lv->flags & 4096))) ? 1 : 0;   // bit 12: force it (parity tests)
  const int gy = pl->pairloop ? Bgrid : Bgrid * npairs(lv);
  int target = (resident + gy - 1) / gy;
  // Mid-size levels (a few tiles per wave at most) start all their waves in the same phase: measured, ~1300 waves
  // finish a tile in 67 us but 2560 need 169 us (160x120 x 8: one tile per wave 21.1 us/window, two per wave on half
  // the waves 16.7).  Such levels run on at most 320 workgroups (1280 waves).
  if (pl->c128 && !pl->patch && (long long)pl->tiles * VBsel < 4LL * kCUs * BANET_G128P_WAVES * kNumWaves)
    target = min(target, (320 + VB - 1) / VB);
  // quarter-tile work items pay (measured: 40x30 x 8 windows 94 -> 50 us) only while they still leave the chip
  // mostly empty -- at most one item per SIMD; beyond that the redone depth dot / geometry costs more than
  // the shorter step chain saves (80x60 x 8: 101 -> 132 us).  flags bit 4: off (A/B).
  pl->qshift = (pl->c128 && !pl->patch && pl->tiles <= 32 && (long long)pl->tiles * VBsel * 4 <= (long long)kCUs * 4 &&
                !(lv->flags & 16)) ? 2 : 0;
  if (pl->c128 && !pl->patch && (lv->flags & 1024)) pl->qshift = 2;   // bit 10: force quarter tiles (A/B)   // (80x60 x 2 windows, also 640 items, LOSES 82 vs 51 us: coarsest levels only)
  int G = pl->c128 ? ((pl->tiles << pl->qshift) + 3) / 4 : pl->groups;
  if (G > target) {
    G = target >= 8 ? (target & ~7) : target;   // several work items per wave: one resident round, no more
  } else if (G >= 8) {
    G = (G + 7) & ~7;   // every item gets its own wave in ONE round (rounding down left a few items for a second
                        // pass: 2x the latency of a level that is pure latency); surplus workgroups find nothing and exit
  }
  if (G < 1) G = 1;
  pl->G = G;
  pl->nbands = (G & 7) == 0 ? 8 : 1;
  pl->pstride = kGHdr + lv->C;
  pl->rows = pl->c128 ? (pl->tiles << pl->qshift) : G;
  pl->frows = pl->rows > kFoldRows ? (pl->rows + kFoldRows - 1) / kFoldRows : pl->rows;
  const size_t row_bytes = (size_t)VB * pl->pstride * sizeof(float);
  pl->off_fold = align_up(row_bytes * pl->rows, 256);
  pl->off_queue = pl->off_fold + (pl->frows != pl->rows ? align_up(row_bytes * pl->frows, 256) : 0);
  pl->partial_bytes = pl->off_queue + align_up((size_t)VB * 8 * sizeof(int), 256);
  pl->rec_bytes = lv->K > 0 ? align_up((size_t)VB * lv->N * 8 * sizeof(float), 256) : 0;
  return BANET_OK;
}

// sum kFoldRows consecutive partial rows (fixed order) -> one row
__global__ __launch_bounds__(256) void ba_fold_kernel(const float* __restrict__ in, int rows, int stride,
                                                      float* __restrict__ out, int frows, const int32_t* active,
                                                      int active_stride, int pairs) {
  const int b = blockIdx.y, c = blockIdx.x, e = threadIdx.x;   // b = virtual window
  if (active != nullptr && active[(size_t)(b / pairs) * active_stride] == 0) return;
  if (e >= stride) return;
  const int r0 = c * kFoldRows, r1 = min(rows, r0 + kFoldRows);
  const float* p = in + ((size_t)b * rows + r0) * stride + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = 0;
  const int n = r1 - r0;
  for (; i + 3 < n; i += 4) {
    s0 += p[(size_t)(i + 0) * stride];
    s1 += p[(size_t)(i + 1) * stride];
    s2 += p[(size_t)(i + 2) * stride];
    s3 += p[(size_t)(i + 3) * stride];
  }
  for (; i < n; ++i) s0 += p[(size_t)i * stride];
  out[((size_t)b * frows + c) * stride + e] = (s0 + s1) + (s2 + s3);
}

void prepare_gather(const banet_level_t* lv, const GatherPlan& pl, float* partials, hipStream_t s) {
  // zeroed by a kernel, not hipMemsetAsync: a captured HIP graph holding the memset node faulted on its second replay
  // (ROCm 7.2; tests/test_gpu_parity.py::test_solve_is_hip_graph_capturable)
  if (pl.c128) launch_zero_iters(reinterpret_cast<int32_t*>(reinterpret_cast<char*>(partials) + pl.off_queue), lv->B * npairs(lv) * 8, s);
}

const float* finish_gather(const banet_level_t* lv, const GatherPlan& pl, const int32_t* active, int active_stride,
                           float* partials, hipStream_t s) {
  if (pl.frows == pl.rows) return partials;
  float* out = reinterpret_cast<float*>(reinterpret_cast<char*>(partials) + pl.off_fold);
  hipLaunchKernelGGL(ba_fold_kernel, dim3(pl.frows, lv->B * npairs(lv)), dim3(256), 0, s, partials, pl.rows, pl.pstride, out,
                     pl.frows, active, active_stride, npairs(lv));
  return out;
}

template <int VEC, int CH, bool GRAD>
static int launch_k(const GatherArgs& a, int K, hipStream_t s) {
  dim3 grid(a.G, a.lv.B * a.pairs), block(kBlock);
  const bool keven = (K & 1) == 0;
  if (K == 0)
    hipLaunchKernelGGL((ba_gather_kernel<VEC, CH, GRAD, 1, 0>), grid, block, 0, s, a);
  else if (keven && K <= 128)
    hipLaunchKernelGGL((ba_gather_kernel<VEC, CH, GRAD, 2, 1>), grid, block, 0, s, a);
  else if (keven && K <= 256)
    hipLaunchKernelGGL((ba_gather_kernel<VEC, CH, GRAD, 2, 2>), grid, block, 0, s, a);
  else if (K <= 64)
    hipLaunchKernelGGL((ba_gather_kernel<VEC, CH, GRAD, 1, 1>), grid, block, 0, s, a);
  else if (K <= 128)
    hipLaunchKernelGGL((ba_gather_kernel<VEC, CH, GRAD, 1, 2>), grid, block, 0, s, a);
  else
    return BANET_ERR_UNSUPPORTED;
  return BANET_OK;
}

template <bool GRAD>
static int launch_c(const GatherArgs& a, int C, int K, hipStream_t s) {
  const bool even = (C & 1) == 0;
  if (even && C <= 128) return launch_k<2, 1, GRAD>(a, K, s);
  if (even && C <= 256) return launch_k<2, 2, GRAD>(a, K, s);
  if (C <= 64) return launch_k<1, 1, GRAD>(a, K, s);
  if (C <= 128) return launch_k<1, 2, GRAD>(a, K, s);
  return BANET_ERR_UNSUPPORTED;
}

int launch_gather(const banet_level_t* lv, const GatherPlan& pl, const float* R, const float* T, const float* Wc,
                  const int32_t* active, int active_stride, float* rec, float* partials, hipStream_t s, unsigned char* mask_out) {
  GatherArgs a;
  a.lv = *lv;
  a.R = R;
  a.T = T;
  a.Wc = Wc;
  a.active = active;
  a.active_stride = active_stride;
  a.rec = rec;
  a.partials = partials;
  a.G = pl.G;
  a.tiles = pl.tiles;
  a.tile_pts = pl.tile_pts;
  a.tiles_x = pl.tiles_x;
  a.tiles_y = pl.tiles_y;
  a.groups = pl.groups;
  a.queue = reinterpret_cast<int*>(reinterpret_cast<char*>(partials) + pl.off_queue);
  a.nbands = pl.nbands;
  a.pairs = npairs(lv);
  a.qshift = pl.qshift;
  a.pairloop = pl.pairloop;
  a.seg_h = pl.strip;
  a.strip_fp = pl.strip_fp;
  a.mask_out = mask_out;
  int rc;
  if (pl.c128)
    rc = pl.quad ? launch_gather128q(a, lv->K, s) : pl.strip ? launch_gather128s(a, lv->K, s) : pl.patch ? launch_gather128p(a, lv->K, s) : launch_gather128(a, lv->K, s);
  else
    rc = lv->tgt_has_grad ? launch_c<true>(a, lv->C, lv->K, s) : launch_c<false>(a, lv->C, lv->K, s);
  if (rc != BANET_OK) return rc;
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
