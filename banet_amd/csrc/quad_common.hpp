// Device helpers shared by the quad-lane C = 128 gather kernels (gather128s.hip: strip segments with a rolling LDS window;
// gather128q.hip: 16-pixel items with direct taps): 4 lanes per pixel, 8 channels per lane and 32-channel slice.
#pragma once
#include "gather_common.hpp"

namespace banet {

// per-chunk state of a segment (one register per chunk of 64 pixels) is held in groups of four chunks: inside a group the
// chunk index is a run-time loop counter (4-way select chains, ~3 instructions per access), across groups a compile-time
// one (the group loop is unrolled) -- an 8-wide register vector with a run-time index costs 7-8 v_cndmask per access, which
// made 32-row segments slower than 16-row ones although they fetch less (profiles/r03_run11_*)
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef int ivec4 __attribute__((ext_vector_type(4)));
typedef float v2fs __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int brev5s(int t) { return (int)(__brev((unsigned)t) >> 27); }

template <int NL, int S0>
__device__ __forceinline__ void carry_push_s(float (&pend)[NL + 1], float v, int t) {
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

// channel maths of one pixel's 4-channel slice (packed fp32), accumulated into q[5] / absd[4]  (= tap_math_p)
__device__ __forceinline__ void tap_math_s(const float4& f1, const float4& a0, const float4& a1, const float4& a2,
                                           const float4& a3, const float4& b0, const float4& b1, const float4& b2,
                                           const float4& b3, const float4& m1, const float4& m2, const float4& p1,
                                           const float4& p2, float w00, float w01, float w10, float w11, float mk,
                                           float (&q)[5], float (&absd)[4]) {
  const float h00 = 0.5f * w00, h01 = 0.5f * w01, h10 = 0.5f * w10, h11 = 0.5f * w11;
#ifdef BANET_TAP_SCALAR   // development A/B (build with -fno-slp-vectorize): the same arithmetic, same order, one channel per instruction
  {
    const float F1[4] = {f1.x, f1.y, f1.z, f1.w}, A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w},
                A2[4] = {a2.x, a2.y, a2.z, a2.w}, A3[4] = {a3.x, a3.y, a3.z, a3.w}, B0[4] = {b0.x, b0.y, b0.z, b0.w},
                B1[4] = {b1.x, b1.y, b1.z, b1.w}, B2[4] = {b2.x, b2.y, b2.z, b2.w}, B3[4] = {b3.x, b3.y, b3.z, b3.w},
                M1[4] = {m1.x, m1.y, m1.z, m1.w}, M2[4] = {m2.x, m2.y, m2.z, m2.w}, P1[4] = {p1.x, p1.y, p1.z, p1.w},
                P2[4] = {p2.x, p2.y, p2.z, p2.w};
    float s11[4], s12[4], s22[4], sg1[4], sg2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float f = ((A1[c] * w00 + A2[c] * w01) + B1[c] * w10) + B2[c] * w11;
      const float gx = (((A2[c] - A0[c]) * h00 + (A3[c] - A1[c]) * h01) + (B2[c] - B0[c]) * h10) + (B3[c] - B1[c]) * h11;
      const float gy = (((B1[c] - M1[c]) * h00 + (B2[c] - M2[c]) * h01) + (P1[c] - A1[c]) * h10) + (P2[c] - A2[c]) * h11;
      const float d = f - F1[c] * mk;
      s11[c] = gx * gx, s12[c] = gx * gy, s22[c] = gy * gy, sg1[c] = gx * d, sg2[c] = gy * d;
      absd[c] += fabsf(d);
    }
    // the packed form's order: lanes {x, z} and {y, w} accumulate separately, then x + y
    q[0] += (s11[0] + s11[2]) + (s11[1] + s11[3]);
    q[1] += (s12[0] + s12[2]) + (s12[1] + s12[3]);
    q[2] += (s22[0] + s22[2]) + (s22[1] + s22[3]);
    q[3] += (sg1[0] + sg1[2]) + (sg1[1] + sg1[3]);
    q[4] += (sg2[0] + sg2[2]) + (sg2[1] + sg2[3]);
    return;
  }
#endif
  v2fs qm11 = {0.f, 0.f}, qm12 = {0.f, 0.f}, qm22 = {0.f, 0.f}, qg1 = {0.f, 0.f}, qg2 = {0.f, 0.f};
#define BANET_V2S(v, k) (v2fs){(k) ? (v).z : (v).x, (k) ? (v).w : (v).y}
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const v2fs F1 = BANET_V2S(f1, k);
    const v2fs A0 = BANET_V2S(a0, k), A1 = BANET_V2S(a1, k), A2 = BANET_V2S(a2, k), A3 = BANET_V2S(a3, k);
    const v2fs B0 = BANET_V2S(b0, k), B1 = BANET_V2S(b1, k), B2 = BANET_V2S(b2, k), B3 = BANET_V2S(b3, k);
    const v2fs M1 = BANET_V2S(m1, k), M2 = BANET_V2S(m2, k), P1 = BANET_V2S(p1, k), P2 = BANET_V2S(p2, k);
    const v2fs f = ((A1 * w00 + A2 * w01) + B1 * w10) + B2 * w11;
    const v2fs gx = (((A2 - A0) * h00 + (A3 - A1) * h01) + (B2 - B0) * h10) + (B3 - B1) * h11;
    const v2fs gy = (((B1 - M1) * h00 + (B2 - M2) * h01) + (P1 - A1) * h10) + (P2 - A2) * h11;
    const v2fs d = f - F1 * mk;
    qm11 += gx * gx;
    qm12 += gx * gy;
    qm22 += gy * gy;
    qg1 += gx * d;
    qg2 += gy * d;
    absd[2 * k] += fabsf(d.x);
    absd[2 * k + 1] += fabsf(d.y);
  }
#undef BANET_V2S
  q[0] += qm11.x + qm11.y;
  q[1] += qm12.x + qm12.y;
  q[2] += qm22.x + qm22.y;
  q[3] += qg1.x + qg1.y;
  q[4] += qg2.x + qg2.y;
}

// sum over the 4 lanes of a quad; every lane of the quad gets the total, fixed order
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<kDppXor1>(v);
  v += dpp_mov<kDppXor2>(v);
  return v;
}
// lane k of every quad -> all four lanes of the quad
template <int K4>
__device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, K4 * 0x55, 0xF, 0xF, true);   // quad_perm:[k,k,k,k]
}
template <int K4>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, quad_bcast<K4>(__builtin_bit_cast(int, v)));
}

struct SGeo {
  float dx, dy, jd0, jd1;
  float jc[12];
  int x0, y0, flags;   // flags: 1 = in the mask, 2 = fast (stencil inside the image), 4 = in the mask but on the rim
};

// the pixel's projection, tap fractions and Jacobian rows from (pixel, D, R, T): statement for statement the geometry
// phase of ba_gather128_kernel (gather128.hip)
// The pose and the window's full-resolution intrinsics as 16 wave-uniform scalars, read ONCE per (window, target frame) by the caller.
// Read inside strip_geometry they were vector loads (the compiler cannot prove that the kernel's own stores do not alias R / T / intr,
// so it neither hoists them nor uses scalar loads): five loads and three s_waitcnt vmcnt(0) round trips per chunk and phase, 24 per
// segment, each of which also drains whatever the wave has in flight (round 5).
struct PoseIntr {
  float R[9], T[3], fx0, fy0, ox0, oy0;
};
__device__ __forceinline__ PoseIntr load_pose_intr(const banet_level_t& lv, int b, const float* __restrict__ Rm, const float* __restrict__ Tv) {
  auto u = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };   // -> SGPRs
  PoseIntr q;
#pragma unroll
  for (int i = 0; i < 9; ++i) q.R[i] = u(Rm[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) q.T[i] = u(Tv[i]);
  q.fx0 = u(lv.intr[b * 4 + 0]);
  q.fy0 = u(lv.intr[b * 4 + 1]);
  q.ox0 = u(lv.intr[b * 4 + 2]);
  q.oy0 = u(lv.intr[b * 4 + 3]);
  return q;
}
__device__ __forceinline__ void strip_geometry(const banet_level_t& lv, const PoseIntr& pq, bool valid, int px, int py, float D, SGeo& o) {
  const int W = lv.W, H = lv.H;
  const float* Rm = pq.R;
  const float* Tv = pq.T;
  o.dx = o.dy = o.jd0 = o.jd1 = 0.f;
  float p0 = 0.f, p1 = 0.f, p2 = 1.f, fx = 1.f, fy = 1.f, ox = 0.f, oy = 0.f;
  if (valid) {
    const float fx0 = pq.fx0, fy0 = pq.fy0, ox0 = pq.ox0, oy0 = pq.oy0;
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
  }
  const float rx = Rm[0] * p0 + Rm[1] * p1 + Rm[2] * p2;
  const float ry = Rm[3] * p0 + Rm[4] * p1 + Rm[5] * p2;
  const float rz = Rm[6] * p0 + Rm[7] * p1 + Rm[8] * p2;
  const float X = rx * D + Tv[0], Y = ry * D + Tv[1], Z = rz * D + Tv[2];
  const float x = X / Z, y = Y / Z;
  const float pxl = fx * x + ox, pyl = fy * y + oy;
  const bool m = valid && (pxl >= 0.f) && (pxl <= (float)(W - 1)) && (pyl >= 0.f) && (pyl <= (float)(H - 1));
#pragma unroll
  for (int i = 0; i < 12; ++i) o.jc[i] = 0.f;
  int x0 = 0, y0 = 0;
  if (m) {
    const float xf = floorf(pxl), yf = floorf(pyl);
    o.dx = pxl - xf;
    o.dy = pyl - yf;
    x0 = (int)xf;
    y0 = (int)yf;
    const float iz = 1.f / Z;
    o.jc[0] = fx * (x * y);
    o.jc[1] = fx * (-1.f - x * x);
    o.jc[2] = fx * y;
    o.jc[3] = fx * (-iz);
    o.jc[4] = 0.f;
    o.jc[5] = fx * (x / Z);
    o.jc[6] = fy * (1.f + y * y);
    o.jc[7] = fy * (-(x * y));
    o.jc[8] = fy * (-x);
    o.jc[9] = 0.f;
    o.jc[10] = fy * (-iz);
    o.jc[11] = fy * (y / Z);
    o.jd0 = fx * ((rx - rz * x) / Z);
    o.jd1 = fy * ((ry - rz * y) / Z);
  }
  const bool interior = (x0 >= 1) && (x0 + 2 <= W - 1) && (y0 >= 1) && (y0 + 2 <= H - 1);
  const bool fast = m && interior;
  o.flags = (m ? 1 : 0) | (fast ? 2 : 0) | ((m && !fast) ? 4 : 0);
  o.x0 = x0;
  o.y0 = y0;
}

}  // namespace banet
