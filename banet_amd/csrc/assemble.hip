// Fused BA assembly for gfx950:  warp -> bilinear sample (+ on-the-fly gradient) ->
// residual -> channel reductions -> Jacobians -> normal equations, one pass over HBM.
//
// Restates, per pixel n of window b (citations under /root/reference):
//   D = D0 + Bs.W                         bundlenet.py:208
//   X = (R p) D + T ; x,y,Z ; px,py       bundlenet.py:209-224 / legacy/ba.py:239-251
//   sample [f|gx|gy] at (px,py), mask     utils_python.py:61-117 / bundlenet.py:154-157
//   d = m (F2w - F1) ; G = m [gx,gy]      legacy/ba.py:258-264 (bundlenet: opposite sign)
//   Jc (2x6), jd (2)                      legacy/ba.py:36-48 ; bundlenet.py:49-74
//   AtA += J^T (G^T G) J ; Atb += J^T G^T d   utils.cu:331-414 == legacy/ba.py:282-283
// with J = [Jc | jd (x) b_n] never materialised:  the 6x6 block and 6xK border are VALU
// work on per-pixel 2x2 / 2-vectors, and only H_dd = sum_n s_n b_n b_n^T (s_n = jd^T M jd)
// goes to the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Work decomposition: grid (G, B); a 256-thread workgroup walks 64-pixel tiles (8x8
// patches in dense mode).  Per tile:
//   P0  basis tile -> LDS (coalesced rows)            P1  D-dot from LDS (skewed, conflict free)
//   P2  geometry, lane = pixel (wave 0)               P3  gather: one pixel per wave at a time,
//       lane = channel pair (512-B coalesced rows), 5 channel sums per pixel reduced with a
//       transposing butterfly (about one shuffle per value), then 6x6 algebra, lane = pixel
//   P4  H_cd / Atb_d rank-1 updates, lane = coefficient pair
//   P5  H_dd on MFMA from the LDS basis tile, upper-triangular 16x16 blocks only
// Every workgroup writes one partial [P*P + P + C + 1]; reduce.hip sums them in fixed order.
#include <vector>

#include "kernels.hpp"

namespace banet {

struct AsmArgs {
  banet_level_t lv;
  const float* R;
  const float* T;
  const float* Wc;
  const int32_t* active;  // optional per-window flag (stride active_stride int32); nullptr = all active
  int active_stride;
  float* partials;        // [B][G][pstride]
  int G;
  int tiles, tiles_x, tiles_y;
  int P;
  int pstride;
};

template <int VEC>
struct Vec {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Vec<VEC> ldv(const float* __restrict__ p, bool ok) {
  Vec<VEC> r;
  if constexpr (VEC == 2) {
    float2 t = make_float2(0.f, 0.f);
    if (ok) t = *reinterpret_cast<const float2*>(p);
    r.v[0] = t.x;
    r.v[1] = t.y;
  } else {
    r.v[0] = ok ? *p : 0.f;
  }
  return r;
}

// dense tile order: vertical strips 8 tiles wide, row-major inside a strip, so that the
// tiles a band of workgroups touches at the same time share target rows in L2.
__device__ __forceinline__ void tile_coords(int t, int tiles_x, int tiles_y, int& tx, int& ty) {
  const int full = tiles_x >> 3;
  const int per_strip = tiles_y << 3;
  if (t < full * per_strip) {
    const int s = t / per_strip;
    const int r = t - s * per_strip;
    ty = r >> 3;
    tx = (s << 3) + (r & 7);
  } else {
    const int r = t - full * per_strip;
    const int wl = tiles_x - (full << 3);
    ty = r / wl;
    tx = (full << 3) + (r - ty * wl);
  }
}

constexpr int kGeoStride = 14;  // main path: x0,y0,w00,w01,w10,w11,flags,point ; [8..13]: true x0,y0,w.. of rim pixels
constexpr int kJStride = 17;    // 12 Jc + 2 jd (+pad: odd stride -> conflict-free per-lane reads)
constexpr int kUStride = 8;     // u0..u5, s, r

// Slow generic path for pixels whose gradient stencil touches the image rim: clamped taps
// (utils_python.py:96-99) and reflect-padded central differences (bundlenet.py:97-99).
// Returns this lane's channel partials (caller reduces across the wave).
template <int VEC, int CH>
__device__ __noinline__ Q5 border_pixel_q5(const float* ge, const float* __restrict__ src_b,
                                           const float* __restrict__ tgt_b, int C, int H, int W, int lane,
                                           float (&absd)[CH][VEC]) {
  Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
  const int x0 = rfl(__float_as_int(ge[8])), y0 = rfl(__float_as_int(ge[9]));
  const float wt[4] = {rfl(ge[10]), rfl(ge[11]), rfl(ge[12]), rfl(ge[13])};
  const int pt = rfl(__float_as_int(ge[7]));
  const int xs[2] = {min(max(x0, 0), W - 1), min(max(x0 + 1, 0), W - 1)};
  const int ys[2] = {min(max(y0, 0), H - 1), min(max(y0 + 1, 0), H - 1)};
  const float* __restrict__ srow = src_b + (size_t)pt * C;
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int c = (ch * 64 + lane) * VEC;
    const bool ok = c < C;
    const Vec<VEC> f1 = ldv<VEC>(srow + c, ok);
    Vec<VEC> f, gx, gy;
#pragma unroll
    for (int e = 0; e < VEC; ++e) f.v[e] = gx.v[e] = gy.v[e] = 0.f;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) {
        const int xc = xs[ix], yc = ys[iy];
        const Vec<VEC> cc = ldv<VEC>(tgt_b + (size_t)(yc * W + xc) * C + c, ok);
        const Vec<VEC> xl = ldv<VEC>(tgt_b + (size_t)(yc * W + refl_m(xc)) * C + c, ok);
        const Vec<VEC> xr = ldv<VEC>(tgt_b + (size_t)(yc * W + refl_p(xc, W)) * C + c, ok);
        const Vec<VEC> yu = ldv<VEC>(tgt_b + (size_t)(refl_m(yc) * W + xc) * C + c, ok);
        const Vec<VEC> yd = ldv<VEC>(tgt_b + (size_t)(refl_p(yc, H) * W + xc) * C + c, ok);
        const float wq = wt[iy * 2 + ix];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          f.v[e] += cc.v[e] * wq;
          gx.v[e] += (0.5f * (xr.v[e] - xl.v[e])) * wq;
          gy.v[e] += (0.5f * (yd.v[e] - yu.v[e])) * wq;
        }
      }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = f.v[e] - f1.v[e];
      q.m11 = fmaf(gx.v[e], gx.v[e], q.m11);
      q.m12 = fmaf(gx.v[e], gy.v[e], q.m12);
      q.m22 = fmaf(gy.v[e], gy.v[e], q.m22);
      q.g1 = fmaf(gx.v[e], d, q.g1);
      q.g2 = fmaf(gy.v[e], d, q.g2);
      absd[ch][e] += fabsf(d);
    }
  }
  return q;
}

template <int NB>
constexpr int lds_floats() {
  constexpr int KPAD = NB * 16;
  constexpr int KP = NB ? KPAD + 16 : 0;
  int n = kTilePix * KP + KPAD + 4 * 64;           // sB, sW, sDp
  n += kTilePix * (kGeoStride + kJStride + kUStride);
  n += 4 * 28 * 16;                                 // sHacc
  // epilogue scratch overlays sB when NB>0; otherwise needs its own room
  const int epi = 4 * 32 + 4 * 256 + 4 * 8 * (KPAD ? KPAD : 4);
  if (NB == 0) n += epi; else if (kTilePix * KP < epi) n += epi - kTilePix * KP;
  return n;
}

template <int NB, int VEC, int CH, bool GRAD>
__global__ __launch_bounds__(kBlock, (NB == 0 ? 4 : 3)) void ba_assemble_kernel(const AsmArgs a) {
  constexpr int KPAD = NB * 16;
  constexpr int KP = NB ? KPAD + 16 : 0;
  constexpr int KQ = NB ? KPAD / 4 : 1;
  constexpr int KV = KPAD >= 64 ? KPAD / 64 : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sB = smem;
  float* sW = sB + kTilePix * KP;
  float* sDp = sW + KPAD;
  float* sGeo = sDp + 4 * 64;
  float* sJ = sGeo + kTilePix * kGeoStride;
  float* sU = sJ + kTilePix * kJStride;
  float* sHacc = sU + kTilePix * kUStride;  // [4 waves][28][16 pixel slots]: H_cc / Atb_c / nvalid partials
  float* sEpi = (NB == 0) ? (sHacc + 4 * 28 * 16) : smem;  // overlays sB after the last tile

  const banet_level_t& lv = a.lv;
  const int b = blockIdx.y;
  const int g = blockIdx.x;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = wave_id();
  const int N = lv.N, C = lv.C, K = lv.K, H = lv.H, W = lv.W;
  const int Ct = GRAD ? 3 * C : C;
  const bool dense = lv.dense != 0;
  const int dbg = lv.reserved_;  // profiling ablation bits (tools/prof_assemble.py); 0 in production

  const float* __restrict__ tgt_b = lv.tgt + (size_t)b * H * W * Ct;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ dep_b = lv.depth + (size_t)b * N;
  const float* __restrict__ bas_b = NB ? lv.basis + (size_t)b * N * K : nullptr;

  if constexpr (NB > 0) {
    for (int k = tid; k < KPAD; k += kBlock) sW[k] = (k < K) ? a.Wc[b * K + k] : 0.f;
  }

  // persistent accumulators -------------------------------------------------------------
  // 21 upper H_cc, 6 Atb_c, 1 nvalid per (wave, pixel slot) live in LDS (ds_add_f32, one
  // owner lane per address -> deterministic) to keep the gather loop's VGPR budget.
  for (int i = tid; i < 4 * 28 * 16; i += kBlock) sHacc[i] = 0.f;
  float absd[CH][VEC];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) absd[i][e] = 0.f;
  float hcd[7][KV];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int e = 0; e < KV; ++e) hcd[i][e] = 0.f;
  // H_dd: wave w owns block rows i1 = w and i2 = NB-1-w of the upper triangle, i.e. NB+1
  // 16x16 blocks: slot q < n1 -> (i1, i1+q), else (i2, i2+q-n1).
  constexpr int NSLOT = NB + 1;
  f32x4 acc[NSLOT];
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int npairs = (NB + 1) / 2;
  const int i1 = w, i2 = NB - 1 - w;
  const bool mf_on = NB > 0 && w < npairs;
  const int n1 = NB - i1;                             // slots of row i1
  const int nslots = (i2 != i1) ? NB + 1 : n1;        // middle row of an odd NB: single row

  // tile schedule: 8 XCD bands when the grid allows it ------------------------------------
  int t_begin, t_end, t_step;
  if ((a.G & 7) == 0) {
    const int x = g & 7, s = g >> 3, per = a.G >> 3;
    t_begin = (int)(((long long)a.tiles * x) >> 3) + s;
    t_end = (int)(((long long)a.tiles * (x + 1)) >> 3);
    t_step = per;
  } else {
    t_begin = g;
    t_end = a.tiles;
    t_step = a.G;
  }

  for (int t = t_begin; t < t_end; t += t_step) {
    int tx = 0, ty = 0;
    if (dense) tile_coords(t, a.tiles_x, a.tiles_y, tx, ty);

    auto point_of = [&](int n, bool& valid) -> int {
      if (dense) {
        const int py = (ty << 3) + (n >> 3), px = (tx << 3) + (n & 7);
        valid = (py < H) && (px < W);
        return py * W + px;
      }
      const int pt = t * kTilePix + n;
      valid = pt < N;
      return pt;
    };

    // ---- P0: basis tile -> LDS -------------------------------------------------------
    if constexpr (NB > 0) {
      if (dbg & 8) {
      } else if ((K & 3) == 0) {
        constexpr int QPR = KPAD / 4;  // float4 per LDS row
        for (int idx = tid; idx < kTilePix * QPR; idx += kBlock) {
          const int n = idx / QPR, q = idx - n * QPR;
          bool valid;
          const int pt = point_of(n, valid);
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid && 4 * q < K) v = *reinterpret_cast<const float4*>(bas_b + (size_t)pt * K + 4 * q);
          *reinterpret_cast<float4*>(sB + n * KP + 4 * q) = v;
        }
      } else {
        for (int idx = tid; idx < kTilePix * KPAD; idx += kBlock) {
          const int n = idx / KPAD, k = idx - n * KPAD;
          bool valid;
          const int pt = point_of(n, valid);
          sB[n * KP + k] = (valid && k < K) ? bas_b[(size_t)pt * K + k] : 0.f;
        }
      }
      __syncthreads();
      // ---- P1: D-dot, wave w covers coefficients [w*KQ, (w+1)*KQ) --------------------
      {
        float acc = 0.f;
        const float* row = sB + lane * KP + w * KQ;
        const float* wv = sW + w * KQ;
#pragma unroll 8
        for (int j = 0; j < KQ; ++j) {
          const int kk = (j + lane) & (KQ - 1);
          acc = fmaf(row[kk], wv[kk], acc);
        }
        sDp[w * 64 + lane] = acc;
      }
      __syncthreads();
    }

    // ---- P2: geometry, lane = pixel --------------------------------------------------
    if (tid < kTilePix) {
      const int n = tid;
      bool valid;
      const int pt = point_of(n, valid);
      // pose / intrinsics are re-read here (scalar loads) instead of living in SGPRs across
      // the whole tile loop
      float Rm[9], Tv[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) Rm[i] = a.R[b * 9 + i];
#pragma unroll
      for (int i = 0; i < 3; ++i) Tv[i] = a.T[b * 3 + i];
      float fx0 = 0.f, fy0 = 0.f, ox0 = 0.f, oy0 = 0.f;
      if (dense) {
        fx0 = lv.intr[b * 4 + 0];
        fy0 = lv.intr[b * 4 + 1];
        ox0 = lv.intr[b * 4 + 2];
        oy0 = lv.intr[b * 4 + 3];
      }
      float D = 0.f, p0 = 0.f, p1 = 0.f, p2 = 1.f, fx = 1.f, fy = 1.f, ox = 0.f, oy = 0.f;
      if (valid) {
        D = dep_b[pt];
        if constexpr (NB > 0) D += (sDp[n] + sDp[64 + n]) + (sDp[128 + n] + sDp[192 + n]);
        if (dense) {
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
      const float rx = Rm[0] * p0 + Rm[1] * p1 + Rm[2] * p2;
      const float ry = Rm[3] * p0 + Rm[4] * p1 + Rm[5] * p2;
      const float rz = Rm[6] * p0 + Rm[7] * p1 + Rm[8] * p2;
      const float X = rx * D + Tv[0], Y = ry * D + Tv[1], Z = rz * D + Tv[2];
      const float x = X / Z, y = Y / Z;
      const float pxl = fx * x + ox, pyl = fy * y + oy;
      const bool m = valid && (pxl >= 0.f) && (pxl <= (float)(W - 1)) && (pyl >= 0.f) && (pyl <= (float)(H - 1));
      int x0 = 0, y0 = 0;
      float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
      float jc[12], jd0 = 0.f, jd1 = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) jc[i] = 0.f;
      if (m) {
        const float xf = floorf(pxl), yf = floorf(pyl);
        const float dx = pxl - xf, dy = pyl - yf;
        x0 = (int)xf;
        y0 = (int)yf;
        w00 = (1.f - dx) * (1.f - dy);
        w01 = dx * (1.f - dy);
        w10 = (1.f - dx) * dy;
        w11 = dx * dy;
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
      // on-the-fly gradient go to the slow path and look masked to the main loop.
      const bool interior = (x0 >= 1) && (x0 + 2 <= W - 1) && (y0 >= 1) && (y0 + 2 <= H - 1);
      const bool fast = m && (GRAD || interior);
      const bool slow = m && !fast;
      float* ge = sGeo + n * kGeoStride;
      ge[0] = __int_as_float(fast ? x0 : 1);
      ge[1] = __int_as_float(fast ? y0 : 1);
      ge[2] = fast ? w00 : 0.f;
      ge[3] = fast ? w01 : 0.f;
      ge[4] = fast ? w10 : 0.f;
      ge[5] = fast ? w11 : 0.f;
      ge[6] = __int_as_float((m ? 1 : 0) | (fast ? 2 : 0) | (slow ? 4 : 0));
      ge[7] = __int_as_float(valid ? pt : 0);
      ge[8] = __int_as_float(x0);
      ge[9] = __int_as_float(y0);
      ge[10] = w00;
      ge[11] = w01;
      ge[12] = w10;
      ge[13] = w11;
      float* jj = sJ + n * kJStride;
#pragma unroll
      for (int i = 0; i < 12; ++i) jj[i] = jc[i];
      jj[12] = jd0;
      jj[13] = jd1;
    }
    __syncthreads();

    // ---- P3: gather + channel reductions; wave w owns pixels 16w..16w+15 -------------
    // Branch-free main path: every pixel issues the same 13 row loads (masked / border
    // pixels read a safe interior location with zero weights), so the loads of the two
    // pixels handled per loop trip overlap.  Border pixels (mask set, stencil touching the
    // image rim) are patched afterwards by the slow generic routine.
    auto pixel_q5 = [&](int n) __attribute__((always_inline)) -> Q5 {
      Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
      const float* ge = sGeo + n * kGeoStride;
      const int x0 = rfl(__float_as_int(ge[0])), y0 = rfl(__float_as_int(ge[1]));
      const float w00 = rfl(ge[2]), w01 = rfl(ge[3]), w10 = rfl(ge[4]), w11 = rfl(ge[5]);
      const int flags = rfl(__float_as_int(ge[6]));
      const int pt = rfl(__float_as_int(ge[7]));
      const float mk = (float)(flags & 1) * (float)((flags >> 1) & 1);  // 1 only on the fast path
      const float* __restrict__ srow = src_b + (size_t)pt * C;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const int c = (ch * 64 + lane) * VEC;
        const bool ok = c < C;
        const Vec<VEC> f1 = ldv<VEC>(srow + c, ok);
        Vec<VEC> f, gx, gy;
        if constexpr (GRAD) {
          const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x0 + 1, 0), W - 1);
          const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y0 + 1, 0), H - 1);
          const float* r00 = tgt_b + (size_t)(y0c * W + x0c) * Ct + c;
          const float* r01 = tgt_b + (size_t)(y0c * W + x1c) * Ct + c;
          const float* r10 = tgt_b + (size_t)(y1c * W + x0c) * Ct + c;
          const float* r11 = tgt_b + (size_t)(y1c * W + x1c) * Ct + c;
          const Vec<VEC> a00 = ldv<VEC>(r00, ok), a01 = ldv<VEC>(r01, ok), a10 = ldv<VEC>(r10, ok), a11 = ldv<VEC>(r11, ok);
          const Vec<VEC> b00 = ldv<VEC>(r00 + C, ok), b01 = ldv<VEC>(r01 + C, ok), b10 = ldv<VEC>(r10 + C, ok), b11 = ldv<VEC>(r11 + C, ok);
          const Vec<VEC> c00 = ldv<VEC>(r00 + 2 * C, ok), c01 = ldv<VEC>(r01 + 2 * C, ok), c10 = ldv<VEC>(r10 + 2 * C, ok), c11 = ldv<VEC>(r11 + 2 * C, ok);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            f.v[e] = ((a00.v[e] * w00 + a01.v[e] * w01) + a10.v[e] * w10) + a11.v[e] * w11;
            gx.v[e] = ((b00.v[e] * w00 + b01.v[e] * w01) + b10.v[e] * w10) + b11.v[e] * w11;
            gy.v[e] = ((c00.v[e] * w00 + c01.v[e] * w01) + c10.v[e] * w10) + c11.v[e] * w11;
          }
        } else {
          // the 4 taps and their +-1 neighbours are 12 distinct texels (interior stencil)
          const float* ra = tgt_b + (size_t)(y0 * W + x0) * C + c;  // row y0, col x0
          const float* rb = ra + (size_t)W * C;                     // row y0+1
          const float* rm = ra - (size_t)W * C;                     // row y0-1
          const float* rp = rb + (size_t)W * C;                     // row y0+2
          const Vec<VEC> a0 = ldv<VEC>(ra - C, ok), a1 = ldv<VEC>(ra, ok), a2 = ldv<VEC>(ra + C, ok), a3 = ldv<VEC>(ra + 2 * C, ok);
          const Vec<VEC> b0 = ldv<VEC>(rb - C, ok), b1 = ldv<VEC>(rb, ok), b2 = ldv<VEC>(rb + C, ok), b3 = ldv<VEC>(rb + 2 * C, ok);
          const Vec<VEC> m1 = ldv<VEC>(rm, ok), m2 = ldv<VEC>(rm + C, ok);
          const Vec<VEC> p1 = ldv<VEC>(rp, ok), p2 = ldv<VEC>(rp + C, ok);
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

    {
      const int base = 16 * w;
      // binary-counter merge tree over 8 pixel pairs (see common.hpp::bfly_slot)
      Q5 p2{0.f, 0.f, 0.f, 0.f, 0.f}, p3 = p2, p4 = p2, q = p2;
      for (int pp = 0; pp < ((dbg & 4) ? 0 : 8); ++pp) {
        const Q5 qa = pixel_q5(base + 2 * pp);
        const Q5 qb = pixel_q5(base + 2 * pp + 1);
        const Q5 m1 = q5_merge(qa, qb, 32);
        if ((pp & 1) == 0) {
          p2 = m1;
        } else {
          const Q5 m2 = q5_merge(p2, m1, 16);
          if ((pp & 2) == 0) {
            p3 = m2;
          } else {
            const Q5 m3 = q5_merge(p3, m2, 8);
            if ((pp & 4) == 0)
              p4 = m3;
            else
              q = q5_merge(p4, m3, 4);
          }
        }
      }
      q = q5_finish(q);
      if constexpr (!GRAD) {
        // patch the pixels whose stencil touches the image rim (rare): slow generic path
        for (int i = 0; i < ((dbg & 1) ? 0 : 16); ++i) {
          const int n = base + i;
          const int flags = rfl(__float_as_int(sGeo[n * kGeoStride + 6]));
          if (flags & 4) {  // wave-uniform
            Q5 e = border_pixel_q5<VEC, CH>(sGeo + n * kGeoStride, src_b, tgt_b, C, H, W, lane, absd);
            e.m11 = wave_sum(e.m11);
            e.m12 = wave_sum(e.m12);
            e.m22 = wave_sum(e.m22);
            e.g1 = wave_sum(e.g1);
            e.g2 = wave_sum(e.g2);
            if (bfly_slot(lane) == i) {
              q.m11 += e.m11;
              q.m12 += e.m12;
              q.m22 += e.m22;
              q.g1 += e.g1;
              q.g2 += e.g2;
            }
          }
        }
      }
      // lane = pixel slot: 6x6 algebra
      const int n = base + bfly_slot(lane);
      const float* jj = sJ + n * kJStride;
      float jc[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) jc[i] = jj[i];
      const float jd0 = jj[12], jd1 = jj[13];
      const float msk = (float)(__float_as_int(sGeo[n * kGeoStride + 6]) & 1);
      float mj[12];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        mj[i] = q.m11 * jc[i] + q.m12 * jc[6 + i];
        mj[6 + i] = q.m12 * jc[i] + q.m22 * jc[6 + i];
      }
      if ((lane & 3) == 0) {
        float* ha = sHacc + (w * 28) * 16 + bfly_slot(lane);
        int o = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) {
            atomicAdd(ha + o * 16, jc[i] * mj[j] + jc[6 + i] * mj[6 + j]);
            ++o;
          }
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(ha + (21 + i) * 16, jc[i] * q.g1 + jc[6 + i] * q.g2);
        atomicAdd(ha + 27 * 16, msk);
      }
      if constexpr (NB > 0) {
        const float md0 = q.m11 * jd0 + q.m12 * jd1, md1 = q.m12 * jd0 + q.m22 * jd1;
        if ((lane & 3) == 0) {
          float* uu = sU + n * kUStride;
#pragma unroll
          for (int i = 0; i < 6; ++i) uu[i] = jc[i] * md0 + jc[6 + i] * md1;
          uu[6] = jd0 * md0 + jd1 * md1;    // s_n
          uu[7] = jd0 * q.g1 + jd1 * q.g2;  // r_n
        }
      }
    }

    if constexpr (NB > 0) {
      __syncthreads();
      // ---- P4: H_cd += u_n b_n^T, Atb_d += r_n b_n ;  lane = coefficient(s) ----------
      {
        const int kb = lane * KV;
        if (kb < KPAD && !(dbg & 2)) {
#pragma unroll 4
          for (int i = 0; i < 16; ++i) {
            const int n = 16 * w + i;
            const float4 ua = *reinterpret_cast<const float4*>(sU + n * kUStride);
            const float4 ub = *reinterpret_cast<const float4*>(sU + n * kUStride + 4);
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
      // ---- P5: H_dd += sum_n s_n b_n b_n^T on the matrix cores -----------------------
      if (mf_on && !(dbg & 16)) {
        const int col = lane & 15, kq = lane >> 4;
#pragma unroll 2
        for (int kk = 0; kk < kTilePix / 4; ++kk) {
          const int pix = 4 * kk + kq;
          const float* row = sB + pix * KP + col;
          const float sv = sU[pix * kUStride + 6];
          const float a1 = sv * row[16 * i1];
          const float a2 = sv * row[16 * i2];
#pragma unroll
          for (int q = 0; q < NSLOT; ++q) {
            if (q < nslots) {  // wave-uniform
              const bool first = q < n1;
              const float bj = row[16 * (first ? i1 + q : i2 + q - n1)];
              acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(first ? a1 : a2, bj, acc[q], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();
  }  // tiles

  // ---- epilogue: one partial per workgroup ---------------------------------------------
  const int P = a.P;
  float* __restrict__ part = a.partials + ((size_t)b * a.G + g) * a.pstride;
  float* sRed = sEpi;             // [4][32]
  float* sAbs = sEpi + 128;       // [4][256]
  float* sH = sEpi + 128 + 1024;  // [4][8][KPAD]
  {
    __syncthreads();  // also orders the sHacc zero-fill when this workgroup had no tile
    // fold the 16 pixel slots of each (wave, quantity): fixed order
    float hsum = 0.f;
    if (tid < 4 * 28) {
      const float* hp = sHacc + tid * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) hsum += hp[i];
    }
    if (tid < 4 * 28) sRed[(tid / 28) * 32 + (tid % 28)] = hsum;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int c = (ch * 64 + lane) * VEC + e;
        if (c < C) sAbs[w * 256 + c] = absd[ch][e];
      }
    if constexpr (NB > 0) {
      const int kb = lane * KV;
      if (kb < KPAD) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
          for (int e = 0; e < KV; ++e) sH[(w * 8 + i) * KPAD + kb + e] = hcd[i][e];
      }
    }
  }
  __syncthreads();
  if (tid < 28) {
    const float v = (sRed[tid] + sRed[32 + tid]) + (sRed[64 + tid] + sRed[96 + tid]);
    if (tid < 21) {
      int i = 0, rem = tid;
      while (rem >= 6 - i) {
        rem -= 6 - i;
        ++i;
      }
      const int j = i + rem;
      part[i * P + j] = v;
      part[j * P + i] = v;
    } else if (tid < 27) {
      part[P * P + (tid - 21)] = v;
    } else {
      part[P * P + P + C] = v;
    }
  }
  for (int c = tid; c < C; c += kBlock)
    part[P * P + P + c] = (sAbs[c] + sAbs[256 + c]) + (sAbs[512 + c] + sAbs[768 + c]);
  if constexpr (NB > 0) {
    // bundle sign convention (bundlenet.py:60,234): J = [-Jc | jd b], d = F1 - F2w
    for (int e = tid; e < 7 * K; e += kBlock) {
      const int i = e / K, k = e - i * K;
      const float v = -((sH[(0 * 8 + i) * KPAD + k] + sH[(1 * 8 + i) * KPAD + k]) +
                        (sH[(2 * 8 + i) * KPAD + k] + sH[(3 * 8 + i) * KPAD + k]));
      if (i < 6) {
        part[i * P + 6 + k] = v;
        part[(6 + k) * P + i] = v;
      } else {
        part[P * P + 6 + k] = v;
      }
    }
    if (mf_on) {
      const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
      for (int q = 0; q < NSLOT; ++q) {
        if (q < nslots) {
          const bool first = q < n1;
          const int bi = first ? i1 : i2, bj = first ? i1 + q : i2 + q - n1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * bi + rq + r, cc = 16 * bj + col;
            if (rr < K && cc < K && (bj > bi || rr <= cc)) {
              part[(6 + rr) * P + 6 + cc] = acc[q][r];
              part[(6 + cc) * P + 6 + rr] = acc[q][r];
            }
          }
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------
// fixed-order reduction of the per-workgroup partials (the deterministic counterpart of
// utils.cu:181-198 ColumnReduceSimpleKernel)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_reduce_kernel(const float* __restrict__ partials, const int32_t* active,
                                                        int active_stride, int G, int pstride, int P, int C,
                                                        float* __restrict__ AtA, float* __restrict__ Atb,
                                                        float* __restrict__ absres, float* __restrict__ nvalid) {
  const int b = blockIdx.y;
  if (active != nullptr && active[(size_t)b * active_stride] == 0) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = P * P + P + (C >= 0 ? C + 1 : 0);  // C < 0: AtA/Atb only
  if (e >= total) return;
  const float* p = partials + (size_t)b * G * pstride + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int gidx = 0;
  for (; gidx + 3 < G; gidx += 4) {
    s0 += p[(size_t)(gidx + 0) * pstride];
    s1 += p[(size_t)(gidx + 1) * pstride];
    s2 += p[(size_t)(gidx + 2) * pstride];
    s3 += p[(size_t)(gidx + 3) * pstride];
  }
  for (; gidx < G; ++gidx) s0 += p[(size_t)gidx * pstride];
  const float v = (s0 + s1) + (s2 + s3);
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
static int nb_for_k(int K) {
  if (K == 0) return 0;
  if (K <= 16) return 1;
  if (K <= 32) return 2;
  if (K <= 64) return 4;
  if (K <= 128) return 8;
  return -1;
}

int plan_assemble(const banet_level_t* lv, AsmPlan* pl) {
  if (!lv || lv->B <= 0 || lv->N <= 0 || lv->C <= 0 || lv->K < 0 || lv->H < 4 || lv->W < 4) return BANET_ERR_INVALID_ARG;
  if (lv->C > 256 || nb_for_k(lv->K) < 0) return BANET_ERR_UNSUPPORTED;
  if (lv->dense && lv->N != lv->H * lv->W) return BANET_ERR_INVALID_ARG;
  pl->nb = nb_for_k(lv->K);
  pl->P = 6 + lv->K;
  if (lv->dense) {
    pl->tiles_x = (lv->W + 7) / 8;
    pl->tiles_y = (lv->H + 7) / 8;
    pl->tiles = pl->tiles_x * pl->tiles_y;
  } else {
    pl->tiles_x = pl->tiles_y = 0;
    pl->tiles = (lv->N + kTilePix - 1) / kTilePix;
  }
  // about 3 resident workgroups per CU on 256 CUs, split across the windows; each
  // workgroup should still see >= 4 tiles so the epilogue amortises.
  int target = (768 + lv->B - 1) / lv->B;
  int G = pl->tiles / 4;
  if (G > target) G = target;
  if (G < 1) G = 1;
  if (G >= 8) G &= ~7;  // enables the 8-band XCD schedule
  pl->G = G;
  pl->pstride = (int)align_up((size_t)pl->P * pl->P + pl->P + lv->C + 1, 4);
  pl->partial_bytes = align_up((size_t)lv->B * G * pl->pstride * sizeof(float), 256);
  return BANET_OK;
}

// ---- optional launch timing ------------------------------------------------------------
namespace {
struct Profiler {
  bool on = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> tag;
  int used = 0;
} g_prof;
}  // namespace

int profile_begin(int max_launches) {
  if (g_prof.on || max_launches <= 0) return BANET_ERR_INVALID_ARG;
  g_prof.ev.resize((size_t)2 * max_launches);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) return BANET_ERR_LAUNCH;
  g_prof.tag.assign(max_launches, 0);
  g_prof.used = 0;
  g_prof.on = true;
  return BANET_OK;
}

int profile_end(int max_tags, int32_t* tag_points, int32_t* tag_launches, double* tag_ms, int32_t* ntags) {
  if (!g_prof.on || !tag_points || !tag_launches || !tag_ms || !ntags) return BANET_ERR_INVALID_ARG;
  int nt = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return BANET_ERR_LAUNCH;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    int k = 0;
    while (k < nt && tag_points[k] != g_prof.tag[i]) ++k;
    if (k == nt) {
      if (nt == max_tags) continue;
      tag_points[nt] = g_prof.tag[i];
      tag_launches[nt] = 0;
      tag_ms[nt] = 0.0;
      ++nt;
    }
    tag_launches[k] += 1;
    tag_ms[k] += ms;
  }
  *ntags = nt;
  for (auto& e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.on = false;
  return BANET_OK;
}

template <int NB>
static int launch_nb(const AsmArgs& a, int C, hipStream_t s) {
  const size_t lds = (size_t)lds_floats<NB>() * sizeof(float);
  dim3 grid(a.G, a.lv.B), block(kBlock);
  const bool even = (C & 1) == 0;
#define BANET_LAUNCH(VEC, CH)                                                                          \
  do {                                                                                                 \
    if (a.lv.tgt_has_grad) {                                                                           \
      auto k = ba_assemble_kernel<NB, VEC, CH, true>;                                                  \
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                   \
    } else {                                                                                           \
      auto k = ba_assemble_kernel<NB, VEC, CH, false>;                                                 \
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                   \
    }                                                                                                  \
  } while (0)
  if (even && C <= 128)
    BANET_LAUNCH(2, 1);
  else if (even && C <= 256)
    BANET_LAUNCH(2, 2);
  else if (C <= 64)
    BANET_LAUNCH(1, 1);
  else if (C <= 128)
    BANET_LAUNCH(1, 2);
  else
    return BANET_ERR_UNSUPPORTED;
#undef BANET_LAUNCH
  return BANET_OK;
}

void launch_reduce(const float* partials, const int32_t* active, int active_stride, int B, int G, int pstride, int P,
                   int C, float* AtA, float* Atb, float* absres, float* nvalid, hipStream_t s) {
  const int total = P * P + P + (C >= 0 ? C + 1 : 0);
  hipLaunchKernelGGL(ba_reduce_kernel, dim3((total + 255) / 256, B), dim3(256), 0, s, partials, active, active_stride,
                     G, pstride, P, C, AtA, Atb, absres, nvalid);
}

int launch_assemble(const banet_level_t* lv, const AsmPlan& pl, const float* R, const float* T, const float* Wc,
                    const int32_t* active, int active_stride, float* partials, float* AtA, float* Atb, float* absres, float* nvalid,
                    hipStream_t s) {
  AsmArgs a;
  a.lv = *lv;
  a.R = R;
  a.T = T;
  a.Wc = Wc;
  a.active = active;
  a.active_stride = active_stride;
  a.partials = partials;
  a.G = pl.G;
  a.tiles = pl.tiles;
  a.tiles_x = pl.tiles_x;
  a.tiles_y = pl.tiles_y;
  a.P = pl.P;
  a.pstride = pl.pstride;
  int rc;
  const bool timed = g_prof.on && (size_t)(2 * g_prof.used + 1) < g_prof.ev.size();
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  switch (pl.nb) {
    case 0: rc = launch_nb<0>(a, lv->C, s); break;
    case 1: rc = launch_nb<1>(a, lv->C, s); break;
    case 2: rc = launch_nb<2>(a, lv->C, s); break;
    case 4: rc = launch_nb<4>(a, lv->C, s); break;
    case 8: rc = launch_nb<8>(a, lv->C, s); break;
    default: rc = BANET_ERR_UNSUPPORTED;
  }
  if (timed) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    g_prof.tag[g_prof.used] = lv->N;
    ++g_prof.used;
  }
  if (rc != BANET_OK) return rc;
  launch_reduce(partials, active, active_stride, lv->B, pl.G, pl.pstride, pl.P, lv->C, AtA, Atb, absres, nvalid, s);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
