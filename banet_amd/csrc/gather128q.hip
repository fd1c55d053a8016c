// ba_gather128q_kernel -- the gather pass for C = 128 on launches that are LATENCY-bound: coarse pyramid levels and small
// batches, where a launch has at most a few work items per resident wave and its time is the serial chain of one item
// (ba_gather128_kernel: a wave walks its 8x8 tile in 16 dependent steps of 4 pixels, 70-150 us per tile whatever the load;
// 40x30 x 32 windows: 71 us for 6 % of the chip's throughput).  Here a work item is a 4x4 block of source pixels and the
// whole block is ONE step:
//
//   * 4 lanes per pixel (lane = 4 p + q), 8 channels per lane and 32-channel slice (as the strip kernel, quad_common.hpp):
//     all 16 pixels of the item are sampled at the same time, taps straight from memory (a 4x4 block's 12-texel stencils
//     overlap in a 7x7 footprint: L1 hits), in 8 units of (slice, 16-byte piece) = 13 loads each with the next unit in flight
//     while one is computed: the chain of an item is depth dot -> geometry -> 8 short units -> algebra, ~4x shorter than a tile's;
//   * the depth dot D = D0 + b . W: the quad's lanes read a pixel's basis row as 16-byte loads (lane q: coefficients
//     16 j + 4 q + e) and reduce in the SAME tree order as the other C = 128 kernels (per-lane fma chain over e, then lane
//     distances 16, 8, 4, 2, 1 of their 32-lane layout), so D -- and with it every projection and mask bit -- is bit-identical;
//   * per-pixel algebra on the quad's lanes (redundantly), lane q = 0 contributes; one partial row per item (28 pose sums
//     through the transposing butterfly, C x sum|d| folded over the 16 pixels in a fixed order): bit-reproducible, and the
//     result does not depend on which wave processed which item;
//   * no LDS staging, no workgroup barrier (a workgroup is 4 independent waves), 2 waves per SIMD with two units of loads in
//     flight (104 registers).  (A form with 1 wave per SIMD and all 8 units = 416 registers in flight was measured: no faster
//     anywhere, slower from 3 items per wave on -- profiles/r04_run5_*: the item's time is not its tap round trips.)
//   * items are assigned statically (wave g of the launch takes items g, g + waves, ...: they cost the same), no atomic queue:
//     at one window the queue's single counter was popped 2 x items times and serialised the launch.
// Same arithmetic per pixel as ba_gather128_kernel; the channel sums are added unit by unit (different rounding order).
// Selected by plan_gather for launches with few items per resident wave (flags bit 25 forces it, bit 30 disables it).
#include "quad_common.hpp"

namespace banet {

constexpr int kC128q = 128;

template <int V>
struct QC {
  static constexpr int value = V;
};

// sum over the 16 lanes that agree in lane & 3 (the 16 pixels of an item, one channel group): fixed order
__device__ __forceinline__ float pixel16_sum(float v) {
  v += dpp_mov<kDppRor8>(v);        // pixels p, p ^ 2 (lanes l, l + 8 of a 16-lane row)
  v += dpp_mov<0x124>(v);           // row_ror:4: + the row's other pixel pair
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

struct QTaps {           // one unit: the 13 rows of 4 channels a pixel's stencil needs
  f32x4 f1, a0, a1, a2, a3, b0, b1, b2, b3, m1, m2, p1, p2;
};
__device__ __forceinline__ float4 f4(const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }

// KV4 = number of 128-coefficient chunks of a basis row (0: pose only; K % 4 == 0, K <= 128 KV4)
template <int KV4>
__global__ __launch_bounds__(kBlock, 2) void ba_gather128q_kernel(const GatherArgs a) {
  __shared__ float sAbs[kNumWaves][kC128q];
  const banet_level_t& lv = a.lv;
  const int vb = blockIdx.y;                   // vb = (window, pair)
  const int b = vb / a.pairs;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int lane = threadIdx.x & 63;
  const int w = wave_id();
  const int N = lv.N, K = lv.K, H = lv.H, W = lv.W;
  constexpr int C = kC128q;
  const float* __restrict__ tgt_b = lv.tgt + (size_t)vb * H * W * C;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ dep_b = lv.depth + (size_t)b * N;
  const float* __restrict__ bas_b = KV4 ? lv.basis + (size_t)b * N * K : nullptr;
  float* __restrict__ rec_b = KV4 ? a.rec + (size_t)vb * N * 8 : nullptr;
  const int nitems = a.tiles, items_x = a.tiles_x;
  float* __restrict__ part_b = a.partials + (size_t)vb * nitems * (kGHdr + C);
  const float* Rm = a.R + vb * 9;
  const float* Tv = a.T + vb * 3;
  const PoseIntr pq = load_pose_intr(lv, b, Rm, Tv);     // 16 scalars, once per kernel: quad_common.hpp
  const int q = lane & 3, p = lane >> 2;
  const int qx = p & 3, qy = p >> 2;
  const int rowC = W * C;

  const int nwaves = gridDim.x * kNumWaves;
  for (int wi = blockIdx.x * kNumWaves + w; wi < nitems; wi += nwaves) {
    const int iy = wi / items_x, ix = wi - iy * items_x;
    const int px = 4 * ix + qx, py = 4 * iy + qy;
    const bool valid = (px < W) && (py < H);
    const int pt = valid ? py * W + px : 0;

    // ---- 1. depth: D = D0 + b . W ------------------------------------------------------------------------------------------
    float D = valid ? dep_b[pt] : 0.f;
    if constexpr (KV4 > 0) {
      // virtual lane li = 4 j + q of the other kernels' 32-lane layout holds coefficients 4 li + e (+ 128 kc)
      const float* row = bas_b + (size_t)pt * K;
      f32x4 bv[KV4][8], wv[KV4][8];
#pragma unroll
      for (int kc = 0; kc < KV4; ++kc)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 128 * kc + 16 * j + 4 * q;
          bv[kc][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + (k < K ? k : 0)));
          wv[kc][j] = *reinterpret_cast<const f32x4*>(a.Wc + (size_t)b * K + (k < K ? k : 0));
        }
      float part[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int kc = 0; kc < KV4; ++kc) {
          const bool in = 128 * kc + 16 * j + 4 * q < K;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = fmaf(bv[kc][j][e], in ? wv[kc][j][e] : 0.f, acc);
        }
        part[j] = acc;
      }
      // lane distances 16, 8, 4 of the 32-lane layout = j ^ 4, j ^ 2, j ^ 1; then 2, 1 = across the quad
      const float t0 = part[0] + part[4], t1 = part[1] + part[5], t2 = part[2] + part[6], t3 = part[3] + part[7];
      const float u0 = t0 + t2, u1 = t1 + t3;
      float tot = u0 + u1;
      tot += dpp_mov<kDppXor2>(tot);
      tot += dpp_mov<kDppXor1>(tot);
      D += tot;
    }

    // ---- 2. geometry (every lane of the quad computes its pixel's) -----------------------------------------------------------
    SGeo ge;
    strip_geometry(lv, pq, valid, px, py, D, ge);
    const bool fast = (ge.flags & 2) != 0;
    const float mk = fast ? 1.f : 0.f;
    const float w00 = mk * ((1.f - ge.dx) * (1.f - ge.dy)), w01 = mk * (ge.dx * (1.f - ge.dy)), w10 = mk * ((1.f - ge.dx) * ge.dy),
                w11 = mk * (ge.dx * ge.dy);
    const int x0 = fast ? ge.x0 : 1, y0 = fast ? ge.y0 : 1;

    // ---- 3. taps: 8 units of (slice s, piece hp) = channels 32 s + 8 q + 4 hp + {0..3}; unit u + 1 in flight under unit u ---
    const float* sp = src_b + (size_t)pt * C + 8 * q;
    const float* ra = tgt_b + (size_t)(y0 * W + x0) * C + 8 * q;
    auto issue = [&](QTaps& t, int u) __attribute__((always_inline)) {
      const int co = 32 * (u >> 1) + 4 * (u & 1);
      const float* r = ra + co;
      t.f1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + co));
      t.a0 = *reinterpret_cast<const f32x4*>(r - C);
      t.a1 = *reinterpret_cast<const f32x4*>(r);
      t.a2 = *reinterpret_cast<const f32x4*>(r + C);
      t.a3 = *reinterpret_cast<const f32x4*>(r + 2 * C);
      t.b0 = *reinterpret_cast<const f32x4*>(r + rowC - C);
      t.b1 = *reinterpret_cast<const f32x4*>(r + rowC);
      t.b2 = *reinterpret_cast<const f32x4*>(r + rowC + C);
      t.b3 = *reinterpret_cast<const f32x4*>(r + rowC + 2 * C);
      t.m1 = *reinterpret_cast<const f32x4*>(r - rowC);
      t.m2 = *reinterpret_cast<const f32x4*>(r - rowC + C);
      t.p1 = *reinterpret_cast<const f32x4*>(r + 2 * rowC);
      t.p2 = *reinterpret_cast<const f32x4*>(r + 2 * rowC + C);
    };
    float qq[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float absd[8][4];   // |d| of this lane's channels, unit by unit (folded over the item's 16 pixels in step 6)
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) absd[u][e] = 0.f;
    auto compute = [&](const QTaps& cur, auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      tap_math_s(f4(cur.f1), f4(cur.a0), f4(cur.a1), f4(cur.a2), f4(cur.a3), f4(cur.b0), f4(cur.b1), f4(cur.b2), f4(cur.b3),
                 f4(cur.m1), f4(cur.m2), f4(cur.p1), f4(cur.p2), w00, w01, w10, w11, mk, qq, absd[u]);
    };
    {
      // two register buffers, unit u + 1 in flight while unit u is computed
      QTaps t0, t1;
      issue(t0, 0);
      auto unit = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        QTaps& cur = (u & 1) ? t1 : t0;
        QTaps& nxt = (u & 1) ? t0 : t1;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + 1 < 8) issue(nxt, u + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(cur, uc);
      };
      unit(QC<0>{});
      unit(QC<1>{});
      unit(QC<2>{});
      unit(QC<3>{});
      unit(QC<4>{});
      unit(QC<5>{});
      unit(QC<6>{});
      unit(QC<7>{});
    }
    Q5 qv;
    qv.m11 = quad_sum(qq[0]);
    qv.m12 = quad_sum(qq[1]);
    qv.m22 = quad_sum(qq[2]);
    qv.g1 = quad_sum(qq[3]);
    qv.g2 = quad_sum(qq[4]);

    // ---- 4. pixels whose stencil touches the image rim (rare): the generic slow routine, one pixel at a time ------------------
    float absd2[1][2] = {{0.f, 0.f}};   // channels 2 lane, 2 lane + 1
    {
      unsigned long long slow = __ballot((ge.flags & 4) != 0 && q == 0);
      while (slow) {   // wave-uniform
        const int j = __builtin_ctzll(slow);
        slow &= slow - 1;
        const float jdx = rdl(ge.dx, j), jdy = rdl(ge.dy, j);
        Q5 e = border_pixel_q5<2, 1>(rdl(ge.x0, j), rdl(ge.y0, j), (1.f - jdx) * (1.f - jdy), jdx * (1.f - jdy), (1.f - jdx) * jdy,
                                     jdx * jdy, src_b + (size_t)rdl(pt, j) * C, tgt_b, C, H, W, lane, absd2);
        e.m11 = wave_sum(e.m11);
        e.m12 = wave_sum(e.m12);
        e.m22 = wave_sum(e.m22);
        e.g1 = wave_sum(e.g1);
        e.g2 = wave_sum(e.g2);
        if ((lane >> 2) == (j >> 2)) {
          qv.m11 += e.m11;
          qv.m12 += e.m12;
          qv.m22 += e.m22;
          qv.g1 += e.g1;
          qv.g2 += e.g2;
        }
      }
    }

    // ---- 5. per-pixel 6x6 algebra; lane q = 0 of a quad contributes; records; the item's 28 pose sums ---------------------------
    float* __restrict__ part = part_b + (size_t)wi * (kGHdr + C);
    const bool own = q == 0;
    {
      const float* jc = ge.jc;
      float mj[12];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        mj[i] = qv.m11 * jc[i] + qv.m12 * jc[6 + i];
        mj[6 + i] = qv.m12 * jc[i] + qv.m22 * jc[6 + i];
      }
      float pend[6];
      int o = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jj = i; jj < 6; ++jj) {
          const float v = jc[i] * mj[jj] + jc[6 + i] * mj[6 + jj];
          carry_push_s<5, 32>(pend, own ? v : 0.f, o);
          ++o;
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) carry_push_s<5, 32>(pend, own ? jc[i] * qv.g1 + jc[6 + i] * qv.g2 : 0.f, 21 + i);
      carry_push_s<5, 32>(pend, own ? (float)(ge.flags & 1) : 0.f, 27);
#pragma unroll
      for (int i = 28; i < 32; ++i) carry_push_s<5, 32>(pend, 0.f, i);
      float tot = pend[5];
      tot += dpp_mov<kDppXor1>(tot);
      const int leaf = brev5s(lane >> 1);
      if ((lane & 1) == 0 && leaf < 28) part[leaf] = tot;
      if (a.mask_out != nullptr && valid && own) a.mask_out[(size_t)vb * N + pt] = (unsigned char)(ge.flags & 1);
      if constexpr (KV4 > 0) {
        if (valid && own) {
          const float md0 = qv.m11 * ge.jd0 + qv.m12 * ge.jd1, md1 = qv.m12 * ge.jd0 + qv.m22 * ge.jd1;
          float4 ua, ub;
          ua.x = jc[0] * md0 + jc[6] * md1;
          ua.y = jc[1] * md0 + jc[7] * md1;
          ua.z = jc[2] * md0 + jc[8] * md1;
          ua.w = jc[3] * md0 + jc[9] * md1;
          ub.x = jc[4] * md0 + jc[10] * md1;
          ub.y = jc[5] * md0 + jc[11] * md1;
          ub.z = ge.jd0 * md0 + ge.jd1 * md1;      // s_n
          ub.w = ge.jd0 * qv.g1 + ge.jd1 * qv.g2;  // r_n
          float4* rp = reinterpret_cast<float4*>(rec_b + (size_t)pt * 8);
          rp[0] = ua;
          rp[1] = ub;
        }
      }
    }

    // ---- 6. the item's C x sum|d|: over the 16 pixels (fixed order), lanes 0..3 publish through LDS, + the rim pixels' ------------
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float tot = pixel16_sum(absd[u][e]);
        if (lane < 4) sAbs[w][32 * (u >> 1) + 8 * q + 4 * (u & 1) + e] = tot;
      }
    sAbs[w][2 * lane] += absd2[0][0];        // same wave: LDS operations retire in program order
    sAbs[w][2 * lane + 1] += absd2[0][1];
    part[kGHdr + lane] = sAbs[w][lane];
    part[kGHdr + 64 + lane] = sAbs[w][64 + lane];
  }  // items
}

int launch_gather128q(const GatherArgs& a, int K, hipStream_t s) {
  dim3 grid(a.G, a.lv.B * a.pairs), block(kBlock);
  if (K == 0)
    hipLaunchKernelGGL((ba_gather128q_kernel<0>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 128)
    hipLaunchKernelGGL((ba_gather128q_kernel<1>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 256)
    hipLaunchKernelGGL((ba_gather128q_kernel<2>), grid, block, 0, s, a);
  else
    return BANET_ERR_UNSUPPORTED;
  return BANET_OK;
}

}  // namespace banet
