// Shared device helpers of the gather kernels (gather.hip: generic C / 3C-map layout;
// gather128.hip: C = 128 fast path).
#pragma once
#include "kernels.hpp"

namespace banet {

struct GatherArgs {
  banet_level_t lv;
  const float* R;
  const float* T;
  const float* Wc;
  const int32_t* active;
  int active_stride;
  float* rec;       // [B][N][8]  u0..u5, s, r   (bundle only)
  float* partials;  // [B][G][kGHdr + C]
  int G, tiles, tiles_x, tiles_y, groups;
  int* queue;       // [B][8] tile-queue heads (ba_gather128_kernel), zero at launch
  int nbands;
  int pairs;        // target frames per window: blockIdx.y = window * pairs + pair
  int pairloop;     // ba_gather128p_kernel: 1 = grid y = window, the window's target frames are looped over inside a tile
  int qshift;       // ba_gather128_kernel: 0 = a work item is a tile, 2 = a quarter tile (small levels)
  int seg_h;        // ba_gather128s_kernel: pixel rows per strip segment (32 or 16)
  unsigned char* mask_out;   // optional (parity diagnostics): [B * pairs][N] the in-image mask bit of every pixel, or nullptr
  int strip_fp;     // ba_gather128s_kernel: 1 = frame-parallel workgroups (`pairs` waves per segment, one per target frame)
  int tile_pts;     // ba_gather_kernel, sparse points: points per wave item (64 or 16)
};

template <int VEC>
struct Vec {
  float v[VEC];
};

// Branch-free row load: lanes beyond the row length read element 0 (always valid) and are
// zeroed by a select afterwards.  No exec-mask branch => the compiler can keep every row load
// of a pixel pair in flight together (an `if (ok) load` splits them into wait-separated blocks).
template <int VEC>
__device__ __forceinline__ Vec<VEC> ldv(const float* __restrict__ row, int c, bool ok) {
  Vec<VEC> r;
  const float* p = row + (ok ? c : 0);
  if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    r.v[0] = ok ? t.x : 0.f;
    r.v[1] = ok ? t.y : 0.f;
  } else {
    const float t = *p;
    r.v[0] = ok ? t : 0.f;
  }
  return r;
}

// same, with the non-temporal hint: for rows that are streamed exactly once (basis, source)
template <int VEC>
__device__ __forceinline__ Vec<VEC> ldv_nt(const float* __restrict__ row, int c, bool ok) {
  Vec<VEC> r;
  const float* p = row + (ok ? c : 0);
  if constexpr (VEC == 2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 t = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
    r.v[0] = ok ? t.x : 0.f;
    r.v[1] = ok ? t.y : 0.f;
  } else {
    const float t = __builtin_nontemporal_load(p);
    r.v[0] = ok ? t : 0.f;
  }
  return r;
}

// dense tile order: vertical strips 8 tiles wide, row-major inside a strip, so that tiles that
// are processed at the same time share target rows in L1/L2.
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

__device__ __forceinline__ float rdl(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int rdl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// binary-counter carry chain of the transposing butterfly: leaf t (0..63) in, result out after
// leaf 63; level L merges with lane distance 32>>L, so leaf t ends on lane bitrev6(t).
__device__ __forceinline__ float merge(float a, float b, int s) { return bfly_merge(a, b, s); }
__device__ __forceinline__ Q5 merge(const Q5& a, const Q5& b, int s) { return q5_merge(a, b, s); }

template <typename TT>
__device__ __forceinline__ void carry_push(TT (&pend)[6], TT v, int t, TT& out) {
  bool done = false;
#pragma unroll
  for (int L = 0; L < 6; ++L) {
    if (!done) {
      if (((t >> L) & 1) == 0) {
        pend[L] = v;
        done = true;
      } else {
        v = merge(pend[L], v, 32 >> L);
      }
    }
  }
  if (!done) out = v;
}

__device__ __forceinline__ int brev6(int t) { return (int)(__brev((unsigned)t) >> 26); }

#ifdef BANET_TIMING  // development aid: per-segment cycle counts of the gather loop (tools/prof_assemble.py)
__device__ __forceinline__ unsigned long long tick() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long realtime() {  // 100 MHz, synchronised across the chip
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define BANET_TICK(var) const unsigned long long var = tick()
#define BANET_TACC(acc, a, b) acc += (float)((b) - (a))
#else
#define BANET_TICK(var)
#define BANET_TACC(acc, a, b)
#endif

// Pixel id n (= the lane that owns the pixel) -> position inside the 8x8 patch.  The butterfly
// visits pixel ids in the order brev6(0), brev6(1), ...; mapping id n to the Z-order (Morton)
// position of brev6(n) makes consecutive visits spatial neighbours (they share 8 of their 12
// target texel rows while those are still in L1/L2).
__device__ __forceinline__ void patch_pos(int n, int& px, int& py) {
  const int z = brev6(n);  // visit order of this pixel id
  px = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
  py = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
}

// Slow generic path for pixels whose gradient stencil touches the image rim: clamped taps
// (utils_python.py:96-99) and reflect-padded central differences (bundlenet.py:97-99).
template <int VEC, int CH>
__device__ __noinline__ Q5 border_pixel_q5(int x0, int y0, float w00, float w01, float w10, float w11,
                                           const float* __restrict__ srow, const float* __restrict__ tgt_b, int C,
                                           int H, int W, int lane, float (&absd)[CH][VEC]) {
  Q5 q{0.f, 0.f, 0.f, 0.f, 0.f};
  const float wt[4] = {w00, w01, w10, w11};
  const int xs[2] = {min(max(x0, 0), W - 1), min(max(x0 + 1, 0), W - 1)};
  const int ys[2] = {min(max(y0, 0), H - 1), min(max(y0 + 1, 0), H - 1)};
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int c = (ch * 64 + lane) * VEC;
    const bool ok = c < C;
    const Vec<VEC> f1 = ldv<VEC>(srow, c, ok);
    Vec<VEC> f, gx, gy;
#pragma unroll
    for (int e = 0; e < VEC; ++e) f.v[e] = gx.v[e] = gy.v[e] = 0.f;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) {
        const int xc = xs[ix], yc = ys[iy];
        const Vec<VEC> cc = ldv<VEC>(tgt_b + (size_t)(yc * W + xc) * C, c, ok);
        const Vec<VEC> xl = ldv<VEC>(tgt_b + (size_t)(yc * W + refl_m(xc)) * C, c, ok);
        const Vec<VEC> xr = ldv<VEC>(tgt_b + (size_t)(yc * W + refl_p(xc, W)) * C, c, ok);
        const Vec<VEC> yu = ldv<VEC>(tgt_b + (size_t)(refl_m(yc) * W + xc) * C, c, ok);
        const Vec<VEC> yd = ldv<VEC>(tgt_b + (size_t)(refl_p(yc, H) * W + xc) * C, c, ok);
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

#ifndef BANET_G128P_WAVES
#define BANET_G128P_WAVES 2   // ba_gather128p_kernel: workgroups per CU (its prefetch registers need 256 VGPRs)
#endif
#ifndef BANET_G128_WAVES
#define BANET_G128_WAVES 3   // ba_gather128_kernel: workgroups per CU (= waves per SIMD) of its launch bounds
#endif
int launch_gather128(const GatherArgs& a, int K, hipStream_t s);
int launch_gather128p(const GatherArgs& a, int K, hipStream_t s);   // gather128p.hip: wave-private LDS patches
int launch_gather128s(const GatherArgs& a, int K, hipStream_t s);   // gather128s.hip: strip segments, rolling LDS window
int launch_gather128q(const GatherArgs& a, int K, hipStream_t s);   // gather128q.hip: 4x4-pixel items, one step per item (latency-bound launches)

}  // namespace banet
