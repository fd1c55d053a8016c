// ba_gather128p_kernel -- ba_gather128_kernel (gather128.hip) with wave-private LDS patches for the taps.
// Two consecutive steps (8 pixels = a 4x2 block of the tile) share most of their 12-texel stencils: their
// bounding box in the target map is 7x5 texels for near-unit local scale, against 8 x 12 = 96 tap loads.
// The wave copies that box once per channel half (<= 36 texels x 256 B = 9 KB) into its OWN LDS patch
// (9 coalesced 16-byte loads per lane instead of 24) and serves the taps with ds_read_b128.  Fewer bytes through
// TA/L1 alone did not pay; what pays is that a unit's loads fit in 44 registers, so the NEXT unit's box is
// prefetched while the current one is computed (software pipeline across units, 2 workgroups per CU).
// No workgroup barrier: waves stay independent (the workgroup-synchronous staged kernels in experiments/ lost
// to exactly that).  A step pair whose box does not fit (local scale > 1 by more than a few %, strong rotation)
// takes the direct loads.  A multi-frame window's target frames are looped over inside a tile, so the tile's
// depth D0 + b.W (a read of its basis rows) is computed once per window.
#include "gather_common.hpp"

namespace banet {

constexpr int kC128p = 128;
constexpr int kPatchTexels = 36;
constexpr int kParStrideP = 8;  // src offset, texel offset, w00,w01,w10,w11 (pre-masked), mask, -

__device__ __forceinline__ void morton8p(int n, int& px, int& py) {  // pixel id -> position in the 8x8 patch
  px = (n & 1) | ((n >> 1) & 2) | ((n >> 2) & 4);
  py = ((n >> 1) & 1) | ((n >> 2) & 2) | ((n >> 3) & 4);
}

__device__ __forceinline__ int brev5p(int t) { return (int)(__brev((unsigned)t) >> 27); }

// carry chain of the transposing butterfly over NL levels, first lane distance S0 (cf. carry_push)
template <int NL, int S0>
__device__ __forceinline__ void carry_push_p(float (&pend)[NL + 1], float v, int t) {
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
typedef float v2fp __attribute__((ext_vector_type(2)));
// channel maths of one pixel's 4-channel slice (packed fp32), accumulated into q[5] / absd[4]
__device__ __forceinline__ void tap_math_p(const float4& f1, const float4& a0, const float4& a1, const float4& a2,
                                           const float4& a3, const float4& b0, const float4& b1, const float4& b2,
                                           const float4& b3, const float4& m1, const float4& m2, const float4& p1,
                                           const float4& p2, float w00, float w01, float w10, float w11, float mk,
                                           float (&q)[5], float* absd) {
  const float h00 = 0.5f * w00, h01 = 0.5f * w01, h10 = 0.5f * w10, h11 = 0.5f * w11;
  v2fp qm11 = {0.f, 0.f}, qm12 = {0.f, 0.f}, qm22 = {0.f, 0.f}, qg1 = {0.f, 0.f}, qg2 = {0.f, 0.f};
#define BANET_V2P(v, k) (v2fp){(k) ? (v).z : (v).x, (k) ? (v).w : (v).y}
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const v2fp F1 = BANET_V2P(f1, k);
    const v2fp A0 = BANET_V2P(a0, k), A1 = BANET_V2P(a1, k), A2 = BANET_V2P(a2, k), A3 = BANET_V2P(a3, k);
    const v2fp B0 = BANET_V2P(b0, k), B1 = BANET_V2P(b1, k), B2 = BANET_V2P(b2, k), B3 = BANET_V2P(b3, k);
    const v2fp M1 = BANET_V2P(m1, k), M2 = BANET_V2P(m2, k), P1 = BANET_V2P(p1, k), P2 = BANET_V2P(p2, k);
    const v2fp f = ((A1 * w00 + A2 * w01) + B1 * w10) + B2 * w11;
    const v2fp gx = (((A2 - A0) * h00 + (A3 - A1) * h01) + (B2 - B0) * h10) + (B3 - B1) * h11;
    const v2fp gy = (((B1 - M1) * h00 + (B2 - M2) * h01) + (P1 - A1) * h10) + (P2 - A2) * h11;
    const v2fp d = f - F1 * mk;
    qm11 += gx * gx;
    qm12 += gx * gy;
    qm22 += gy * gy;
    qg1 += gx * d;
    qg2 += gy * d;
    absd[2 * k] += fabsf(d.x);
    absd[2 * k + 1] += fabsf(d.y);
  }
#undef BANET_V2P
  q[0] += qm11.x + qm11.y;
  q[1] += qm12.x + qm12.y;
  q[2] += qm22.x + qm22.y;
  q[3] += qg1.x + qg1.y;
  q[4] += qg2.x + qg2.y;
}

// US = steps (of 4 pixels) per unit: 2 = a 4x2 pixel block, box <= 36 texels, 9 + 2 loads per lane in flight (the product
// kernel of round 1); 4 = a 4x4 block, box <= 52 texels, 13 + 4 loads: half as many load -> LDS -> tap latency events per
// tile, 3.06 instead of 4.4 box texels per pixel through the L1.
// FS = fixed-stride patch: the box is stored with a row stride of 8 texels (boxes up to 8 x 5) and fetched with buffer loads --
// per-lane column offsets (2 per unit half) in voffset, the row offsets in the scalar offset: no per-load address arithmetic
// (the packed layout costs 8 VALU instructions of row / column bookkeeping and 64-bit address maths per load, 96 per unit half),
// and the tap rows of the patch sit at compile-time distances.
template <int KV4, int US, bool FS>
__global__ __launch_bounds__(kBlock, BANET_G128P_WAVES) void ba_gather128p_kernel(const GatherArgs a) {
  constexpr int PT = FS ? 40 : (US == 2 ? kPatchTexels : 52), NL = (PT + 3) / 4, GL = 4 * US;   // patch texels, box loads per lane, lanes per unit
  __shared__ __attribute__((aligned(16))) float sPar[kNumWaves][64][kParStrideP];
  __shared__ float sQ[kNumWaves][64][5];
  __shared__ float sAbs[kNumWaves][kC128p];
  __shared__ __attribute__((aligned(16))) float sPatch[kNumWaves][PT * 64];   // one channel half of a unit's box
  __shared__ int sGrp[kNumWaves][8][4];                                        // per unit: base offset, pw, staged texels, ph
  const banet_level_t& lv = a.lv;
  // pairloop (levels with enough tiles per window): a workgroup serves one window and a multi-frame window's target frames
  // ("pairs") are looped over INSIDE a tile, so the depth D0 + b.W is computed once; otherwise grid y = (window, pair)
  // and a work item is one pair's tile (4x finer items for the levels where items are scarce).
  const int g = blockIdx.x;
  const int b = a.pairloop ? blockIdx.y : blockIdx.y / a.pairs;
  const int pr_lo = a.pairloop ? 0 : blockIdx.y % a.pairs, pr_hi = a.pairloop ? a.pairs : pr_lo + 1;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id();
  const int N = lv.N, K = lv.K, H = lv.H, W = lv.W;
  constexpr int C = kC128p;
  const bool dense = lv.dense != 0;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ dep_b = lv.depth + (size_t)b * N;
  const float* __restrict__ bas_b = KV4 ? lv.basis + (size_t)b * N * K : nullptr;
  const int qshift = a.qshift;                       // 2: a work item is one quarter (16 pixels, 4 steps) of a tile
  const int nitems = a.tiles << qshift;
  const int grp = lane >> 4, sub = lane & 15;
  const int half = lane >> 5, li = lane & 31;

#ifdef BANET_ABLATE   // development aid (tools/prof_assemble.py): flags bits 0..3 switch phases of this kernel off
  const int abl = lv.flags;
#else
  constexpr int abl = 0;
#endif
  float wreg[KV4 ? KV4 : 1][4];  // this lane's slice of the depth coefficients
  // The waves of a workgroup start 7 us apart (s_sleep 127 = 8128 cycles): launched together they stay in lock-step through
  // the first tiles (depth-dot burst, then every unit's loads at the same time); measured 320x240 x 8: 39.0 -> 37.6
  // us/window, 640x480 x 8: 125.5 -> 124.1; 15 us per wave: no better.  (On the short levels, which run the direct
  // kernel, any stagger loses.)
  if (!(lv.flags & 2048))   // bit 11: no stagger (A/B)
    for (int i = 0; i < 2 * w; ++i) __builtin_amdgcn_s_sleep(127);
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
  int* __restrict__ queue = a.queue + blockIdx.y * 8;
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
    int tx = 0, ty = 0;
    if (dense) tile_coords(t, a.tiles_x, a.tiles_y, tx, ty);
    auto point_of = [&](int n, bool& valid) -> int {
      if (dense) {
        int qx, qy;
        morton8p(n, qx, qy);
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
    BANET_TICK(tp0);
#if defined(BANET_TIMING) && BANET_TIMING == 1
    float tp_stage = 0.f, tp_taps = 0.f;
#endif

    // ---- 1. depth: D_j = D0_j + b_j . W.  A half wave reads one basis row per instruction
    // (16 B per lane); 32 row pairs go through a 5-level transposing butterfly inside each half,
    // leaf t of half h carrying pixel h*32 + brev5p(t), so that pixel j's sum lands on lane j.
    float D = valid ? dep_b[pt] : 0.f;
    if (KV4 > 0 && !(abl & 4)) {
      float pend[6];
      float part[32];   // all 32 row loads of the half wave in flight together (2 waves per SIMD: 256 VGPRs)
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        bool vj;
        const int ptj = point_of(half * 32 + brev5p(i), vj);
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
      for (int i = 0; i < 32; ++i) carry_push_p<5, 16>(pend, part[i], i);
      D += pend[5];
    }

#pragma unroll 1
    for (int pr = pr_lo; pr < pr_hi; ++pr) {   // target frames of the window: same pixels, depth and source features
    const int vb = b * a.pairs + pr;
    const float* __restrict__ tgt_b = lv.tgt + (size_t)vb * H * W * C;
    float* __restrict__ rec_b = KV4 ? a.rec + (size_t)vb * N * 8 : nullptr;
    float* __restrict__ part_b = a.partials + (size_t)vb * nitems * (kGHdr + C);
    // |d| of channels {4 sub + e} (absA) and {64 + 4 sub + e} (absB) over this lane group's pixels.  Two arrays that trade places
    // after every channel half instead of one array indexed by the (rolled) half loop: `absd8[4 * h + e]` with a run-time h cost
    // 15 v_cndmask per accumulation (8-way select on read and on write), 116 of the 259 VALU instructions of a unit half.
    float absA[4] = {0.f, 0.f, 0.f, 0.f}, absB[4] = {0.f, 0.f, 0.f, 0.f};
    float absd2[1][2] = {{0.f, 0.f}};  // rim pixels (generic routine: channels 2 lane, 2 lane + 1)

    // ---- 2. geometry, lane = pixel -------------------------------------------------------
    float gw00 = 0.f, gw01 = 0.f, gw10 = 0.f, gw11 = 0.f, jd0 = 0.f, jd1 = 0.f;
    float jc[12];
    int gx0 = 1, gy0 = 1, gflags = 0;
    // the pixel's projection, tap weights and Jacobian rows from (pt, D, R, T)
    auto pixel_geometry = [&]() __attribute__((always_inline)) {
      gw00 = gw01 = gw10 = gw11 = jd0 = jd1 = 0.f;
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
    };
    pixel_geometry();
    {
      const int x0 = gx0, y0 = gy0;
      const bool fast = (gflags & 2) != 0;
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
      // bounding box of the unit's pixel group (lanes GL k .. GL k + GL - 1 = unit k) over its pixels that need taps
      {
        auto red8 = [](int v, bool mx) {
          int o;
          if (US == 4) {
            o = __builtin_amdgcn_update_dpp(v, v, kDppRor8, 0xF, 0xF, false);
            v = mx ? max(v, o) : min(v, o);
          }
          o = __builtin_amdgcn_update_dpp(v, v, kDppHalfMirror, 0xF, 0xF, false);
          v = mx ? max(v, o) : min(v, o);
          o = __builtin_amdgcn_update_dpp(v, v, kDppXor2, 0xF, 0xF, false);
          v = mx ? max(v, o) : min(v, o);
          o = __builtin_amdgcn_update_dpp(v, v, kDppXor1, 0xF, 0xF, false);
          return mx ? max(v, o) : min(v, o);
        };
        const int big = 0x3fffffff;
        const int bx0 = red8(fast ? x0 : big, false), bx1 = red8(fast ? x0 : -big, true);
        const int by0 = red8(fast ? y0 : big, false), by1 = red8(fast ? y0 : -big, true);
        const int x_lo = bx0 - 1, y_lo = by0 - 1, pw = bx1 - bx0 + 4, ph = by1 - by0 + 4;
        const bool st = bx1 >= bx0 && (FS ? (pw <= 8 && ph <= 5) : pw * ph <= PT) && !(lv.flags & 128);
        pb.w = __int_as_float(((fast ? y0 - y_lo : 1) * (FS ? 8 : pw) + (fast ? x0 - x_lo : 1)) * 64);
        if ((lane & (GL - 1)) == 0) {
          sGrp[w][lane / GL][0] = st ? (y_lo * W + x_lo) * C : 0;
          sGrp[w][lane / GL][1] = st ? pw : 0;
          sGrp[w][lane / GL][2] = st ? pw * ph : 0;
          sGrp[w][lane / GL][3] = st ? ph : 0;
        }
      }
      *reinterpret_cast<float4*>(&sPar[w][lane][0]) = pa;
      *reinterpret_cast<float4*>(&sPar[w][lane][4]) = pb;
    }

    BANET_TICK(tp2);
    // ---- 3. gather: 8 step pairs x (2 steps x 4 pixels); lane = (pixel group, 8-channel slice) -----
    // Software pipeline over the units (pair, channel half): the 9 box loads + 2 source loads of the NEXT
    // staged unit are issued (into 44 registers) right after the current unit's box has been written to the
    // wave's LDS patch and before its taps are computed, so their memory latency hides behind ~330 VALU
    // instructions -- the direct kernel would need 104 registers to prefetch one step.
    const int rowC = W * C;
    f32x4 pst[NL], pf1[US];
    bool pre = false;
    // buffer resources of this virtual window's target map and this window's source map (wave-uniform)
    [[maybe_unused]] const auto rs_tgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tgt_b), 0, H * W * C * 4, 0x00020000);
    [[maybe_unused]] const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_b), 0, N * C * 4, 0x00020000);
    auto issue = [&](int sp_, int h_) __attribute__((always_inline)) {
      const int gb = rfl(sGrp[w][sp_][0]), pw_ = rfl(sGrp[w][sp_][1]), ph_ = rfl(sGrp[w][sp_][3]);
      if constexpr (FS) {
        const int qc = lane >> 4;
        const unsigned c0 = (unsigned)(min(qc, pw_ - 1) * (C * 4) + 16 * sub), c1 = (unsigned)(min(qc + 4, pw_ - 1) * (C * 4) + 16 * sub);
        const int sbase = gb * 4 + 256 * h_;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const int srow = sbase + min(i >> 1, ph_ - 1) * (W * C * 4);     // scalar
          pst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_tgt, (i & 1) ? c1 : c0, srow, 0));
        }
#pragma unroll
        for (int t = 0; t < US; ++t) {
          const unsigned osrc = (unsigned)__float_as_int(sPar[w][4 * (US * sp_ + t) + grp][0]);
          pf1[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, osrc * 4u + 16u * (unsigned)sub, 256 * h_, 2));
        }
      } else {
        int row = 0, col = lane >> 4;                       // texel (lane >> 4) + 4 i of the box, pw_ >= 4
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          unsigned off = (unsigned)gb + (unsigned)((min(row, ph_ - 1) * W + col) * C) + 4u * (unsigned)sub + 64u * (unsigned)h_;
          if (abl & 1) off = 4u * (unsigned)sub + 64u * (unsigned)h_ + (unsigned)((lane >> 4) * C);   // every box = texels 0..3
          pst[i] = *reinterpret_cast<const f32x4*>(tgt_b + (size_t)off);
          col += 4;
          if (col >= pw_) {
            col -= pw_;
            row += 1;
          }
        }
#pragma unroll
        for (int t = 0; t < US; ++t) {
          unsigned osrc = (unsigned)__float_as_int(sPar[w][4 * (US * sp_ + t) + grp][0]);
          if (abl & 2) osrc = (unsigned)(grp * C);                                                        // every source row = pixels 0..3
          pf1[t] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src_b + (size_t)(osrc + 64u * (unsigned)h_ + 4u * sub)));
        }
      }
    };
    const int sp_end = s_hi / US;
    // unit order: 0..7 = left / right unit of pixel rows 0-1, 2-3, ... (Z order); flags bit 17 (experiment, US = 2 only):
    // the left column top to bottom, then the right column bottom to top -- horizontal neighbours 1..7 units apart instead of 1
    const bool colmajor = US == 2 && (lv.flags & 131072) != 0;
    auto unit_at = [&](int it) { return colmajor ? (it < 4 ? 2 * it : 2 * (7 - it) + 1) : it; };
    for (int it = s_lo / US; it < sp_end; ++it) {
      const int sp = unit_at(it);
      const int pw = rfl(sGrp[w][sp][1]), ntex = rfl(sGrp[w][sp][2]);
      float qa2[US][5];
#pragma unroll
      for (int t = 0; t < US; ++t)
#pragma unroll
        for (int i = 0; i < 5; ++i) qa2[t][i] = 0.f;
      if (ntex > 0) {   // wave-uniform: the unit's box fits the patch
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
#if BANET_TIMING == 1
          BANET_TICK(tu0);
#endif
          if (!pre) issue(sp, h);
#pragma unroll
          for (int i = 0; i < NL; ++i)
            if (US == 2 || (lane + 64 * i) < PT * 16) *reinterpret_cast<f32x4*>(&sPatch[w][(lane + 64 * i) * 4]) = pst[i];
          f32x4 f1v[US];
#pragma unroll
          for (int t = 0; t < US; ++t) f1v[t] = pf1[t];
          {   // prefetch the next staged unit
            const int nh = h ^ 1, nsp = h ? (it + 1 < sp_end ? unit_at(it + 1) : sp_end) : sp;
            pre = nsp < sp_end && rfl(sGrp[w][nsp][2]) > 0;
            if (pre) issue(nsp, nh);
          }
          const int rs = FS ? 8 * 64 : pw * 64;
#if BANET_TIMING == 1
          BANET_TICK(tu1);   // (the tick waits for the LDS writes of the box)
          BANET_TACC(tp_stage, tu0, tu1);
#endif
          if (!(abl & 8))
#pragma unroll
          for (int t = 0; t < US; ++t) {
            const int j = 4 * (US * sp + t) + grp;
            const float4 pa = *reinterpret_cast<const float4*>(&sPar[w][j][0]);
            const float4 pb = *reinterpret_cast<const float4*>(&sPar[w][j][4]);
            const float* l = &sPatch[w][0] + __float_as_int(pb.w) + 4 * sub;
            const float4 f1 = make_float4(f1v[t][0], f1v[t][1], f1v[t][2], f1v[t][3]);
            const float4 a0 = *reinterpret_cast<const float4*>(l - 64), a1 = *reinterpret_cast<const float4*>(l),
                         a2 = *reinterpret_cast<const float4*>(l + 64), a3 = *reinterpret_cast<const float4*>(l + 128);
            const float4 b0 = *reinterpret_cast<const float4*>(l + rs - 64), b1 = *reinterpret_cast<const float4*>(l + rs),
                         b2 = *reinterpret_cast<const float4*>(l + rs + 64), b3 = *reinterpret_cast<const float4*>(l + rs + 128);
            const float4 m1 = *reinterpret_cast<const float4*>(l - rs), m2 = *reinterpret_cast<const float4*>(l - rs + 64);
            const float4 p1 = *reinterpret_cast<const float4*>(l + 2 * rs), p2 = *reinterpret_cast<const float4*>(l + 2 * rs + 64);
            tap_math_p(f1, a0, a1, a2, a3, b0, b1, b2, b3, m1, m2, p1, p2, pa.z, pa.w, pb.x, pb.y, pb.z, qa2[t], absA);
          }
#if BANET_TIMING == 1
          {
            BANET_TICK(tu2);
            BANET_TACC(tp_taps, tu1, tu2);
          }
#endif
#pragma unroll
          for (int e = 0; e < 4; ++e) {   // half h done: the other half's accumulators become current (two swaps = identity)
            const float tmp = absA[e];
            absA[e] = absB[e];
            absB[e] = tmp;
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < US; ++t) {
          const int j = 4 * (US * sp + t) + grp;
          const float4 pa = *reinterpret_cast<const float4*>(&sPar[w][j][0]);
          const float4 pb = *reinterpret_cast<const float4*>(&sPar[w][j][4]);
          const unsigned osrc = (unsigned)__float_as_int(pa.x), oa = (unsigned)__float_as_int(pa.y);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned co = 64u * h + 4u * sub;
            const float* ra = tgt_b + (size_t)(oa + co);
            const float* rb = ra + rowC;
            const float* rm = ra - rowC;
            const float* rp = rb + rowC;
            const f32x4 f1v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src_b + (size_t)(osrc + co)));
            const float4 f1 = make_float4(f1v[0], f1v[1], f1v[2], f1v[3]);
            const float4 a0 = *reinterpret_cast<const float4*>(ra - C), a1 = *reinterpret_cast<const float4*>(ra),
                         a2 = *reinterpret_cast<const float4*>(ra + C), a3 = *reinterpret_cast<const float4*>(ra + 2 * C);
            const float4 b0 = *reinterpret_cast<const float4*>(rb - C), b1 = *reinterpret_cast<const float4*>(rb),
                         b2 = *reinterpret_cast<const float4*>(rb + C), b3 = *reinterpret_cast<const float4*>(rb + 2 * C);
            const float4 m1 = *reinterpret_cast<const float4*>(rm), m2 = *reinterpret_cast<const float4*>(rm + C);
            const float4 p1 = *reinterpret_cast<const float4*>(rp), p2 = *reinterpret_cast<const float4*>(rp + C);
            tap_math_p(f1, a0, a1, a2, a3, b0, b1, b2, b3, m1, m2, p1, p2, pa.z, pa.w, pb.x, pb.y, pb.z, qa2[t], h ? absB : absA);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < US; ++t) {
        const int j = 4 * (US * sp + t) + grp;
        float qq[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) qq[i] = row16_sum(qa2[t][i]);
        if (sub == 0) {
#pragma unroll
          for (int i = 0; i < 5; ++i) sQ[w][j][i] = qq[i];
        }
      }
    }
    BANET_TICK(tp3);
    Q5 q;
    {
      q.m11 = sQ[w][lane][0];
      q.m12 = sQ[w][lane][1];
      q.m22 = sQ[w][lane][2];
      q.g1 = sQ[w][lane][3];
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
      // 5 levels (lane distance 32..2) + one xor-1 add: lane l ends with leaf brev5p(l >> 1).
      float pend[6];
      int o = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jj = i; jj < 6; ++jj) {
          carry_push_p<5, 32>(pend, jc[i] * mj[jj] + jc[6 + i] * mj[6 + jj], o);
          ++o;
        }
#pragma unroll
      for (int i = 0; i < 6; ++i) carry_push_p<5, 32>(pend, jc[i] * q.g1 + jc[6 + i] * q.g2, 21 + i);
      carry_push_p<5, 32>(pend, (float)(gflags & 1), 27);
      if (a.mask_out != nullptr && valid && mine) a.mask_out[(size_t)vb * N + pt] = (unsigned char)(gflags & 1);
#pragma unroll
      for (int i = 28; i < 32; ++i) carry_push_p<5, 32>(pend, 0.f, i);
      float tot = pend[5];
      tot += dpp_mov<kDppXor1>(tot);
      const int leaf = brev5p(lane >> 1);
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

    BANET_TICK(tp4);
    // ---- 5. the tile's C x sum|d| ------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // fold the 4 pixel groups (fixed order), group 0 publishes
      float v = i < 4 ? absA[i] : absB[i - 4];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (grp == 0) sAbs[w][(i >> 2) * 64 + 4 * sub + (i & 3)] = v;
    }
    sAbs[w][2 * lane] += absd2[0][0];       // same wave: LDS operations retire in program order
    sAbs[w][2 * lane + 1] += absd2[0][1];
    part[kGHdr + lane] = sAbs[w][lane];
    part[kGHdr + 64 + lane] = sAbs[w][64 + lane];
#ifdef BANET_TIMING   // tools/time_gather.py: cycles of this tile (wave-private counters, the last target frame of the window)
    {
      BANET_TICK(tp9);
      if (lane == 0) {
#if BANET_TIMING == 1
        part[28] = tp_stage;                 // unit halves: wait for the box + ds_write + issue of the next unit's loads
        part[29] = tp_taps;                  // unit halves: ds_read of the taps + channel maths
        part[30] = (float)(tp2 - tp0);       // depth dot + geometry
#else                                        // -DBANET_TIMING=2: no ticks inside the unit loop
        part[28] = (float)(tp3 - tp2);       // the unit loop
        part[29] = (float)(tp4 - tp3);       // rim pixels, per-pixel algebra, pose sums, records
        part[30] = (float)(tp9 - tp4);       // sum|d| fold + partial
#endif
        part[31] = (float)(tp9 - tp0);       // whole tile
      }
    }
#endif
    }  // pairs
  }  // tiles
}

int launch_gather128p(const GatherArgs& a, int K, hipStream_t s) {
  dim3 grid(a.G, a.pairloop ? a.lv.B : a.lv.B * a.pairs), block(kBlock);
  // flags bit 13 (A/B experiment, experiments/README.md): 60 KB of unused dynamic LDS per workgroup -> ONE workgroup per CU
  // (one gather wave per SIMD), the occupancy a wave-specialised gather + MFMA-accumulator kernel would leave the gather
  size_t dyn = 0;
  if (a.lv.flags & 8192) {
    dyn = 60 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_gather128p_kernel<1, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  }
  const bool u4 = (a.lv.flags & 16384) != 0;   // bit 14: 4-step units (A/B, experiments/README.md)
  const bool packed = (a.lv.flags & 65536) != 0;   // bit 16: the packed patch with flat loads (A/B)
  const bool small = (size_t)a.lv.H * a.lv.W * a.lv.C * 4 < ((size_t)1 << 31) && (size_t)a.lv.N * a.lv.C * 4 < ((size_t)1 << 31);
  const bool fs = small && !packed && !dyn;          // the fixed-stride patch addresses the maps with 32-bit buffer offsets
  if (K == 0 && fs)
    hipLaunchKernelGGL((ba_gather128p_kernel<0, 2, true>), grid, block, 0, s, a);
  else if (K == 0)
    hipLaunchKernelGGL((ba_gather128p_kernel<0, 2, false>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 128 && u4)
    hipLaunchKernelGGL((ba_gather128p_kernel<1, 4, false>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 128 && (packed || !small || dyn))
    hipLaunchKernelGGL((ba_gather128p_kernel<1, 2, false>), grid, block, dyn, s, a);
  else if ((K & 3) == 0 && K <= 128)
    hipLaunchKernelGGL((ba_gather128p_kernel<1, 2, true>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 256 && fs)
    hipLaunchKernelGGL((ba_gather128p_kernel<2, 2, true>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 256)
    hipLaunchKernelGGL((ba_gather128p_kernel<2, 2, false>), grid, block, 0, s, a);
  else
    return BANET_ERR_UNSUPPORTED;
  return BANET_OK;
}

}  // namespace banet
