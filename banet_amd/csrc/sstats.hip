// Per-pixel sampling statistics and their adjoint -- the C-wide part of a differentiable BA iteration in the reference's
// own tensor layout (bundlenet.py:230-243 forward; the TF autodiff of the same statements backward):
//   samp = resampler(conv2 = [f | gx | gy], (px, py)) ; mask = px in [0, W-1] and py in [0, H-1]        :230-233
//   d = (conv1 - samp_f) mask ; G = [samp_gx, samp_gy] mask                                             :234-239
//   M = G^T G (2x2), g = G^T d (2), sum_n |d| per channel (-> avg residual, :243)
// The training graph (banet_amd/bundlenet.py) needs nothing else that is C wide: J^T (G^T G) J and J^T G^T d follow from
// M, g and the per-pixel Jacobians, so samp / diff / grad / J are never materialised ([B,N,3C] + [B,N,C,3] + [B,N,2,P]
// floats in the reference-style graph).  Backward: given dL/dM, dL/dg (per pixel) and dL/d(sum |d|) (per channel) it
// returns dL/dconv1, dL/dconv2 (scatter-add over the 4 bilinear taps: float atomics, order not reproducible to the
// last bit, as in any scatter-based resampler gradient) and dL/d(px, py) through the bilinear weights (the mask is
// piecewise constant).  Taps outside the image contribute 0 (tf.contrib.resampler semantics).
// One wave per pixel at a time, lane = channel (+64 j): every tap row is a coalesced read of 3C floats.
#include "kernels.hpp"

namespace banet {

namespace {
constexpr int kPixPerWave = 16, kPixPerBlock = kPixPerWave * kNumWaves, kMaxCJ = 4;   // C <= 256

struct Taps {
  unsigned off[4]; // element offset of texel (x0,y0), (x1,y0), (x0,y1), (x1,y1) in conv2 (in floats)
  float w[4];      // bilinear weights
  bool in[4];      // tap inside the image
  float ax, ay;
  bool m;          // mask (bundlenet.py:231-233)
};

__device__ __forceinline__ Taps make_taps(float px, float py, int b, int H, int W, int C3) {
  Taps t;
  t.m = (px >= 0.f) && (px <= (float)(W - 1)) && (py >= 0.f) && (py <= (float)(H - 1));
  const float xf = floorf(t.m ? px : 0.f), yf = floorf(t.m ? py : 0.f);
  const int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
  t.ax = (t.m ? px : 0.f) - xf;
  t.ay = (t.m ? py : 0.f) - yf;
  t.w[0] = (1.f - t.ax) * (1.f - t.ay);
  t.w[1] = t.ax * (1.f - t.ay);
  t.w[2] = (1.f - t.ax) * t.ay;
  t.w[3] = t.ax * t.ay;
  const bool xi = x1 <= W - 1, yi = y1 <= H - 1;
  t.in[0] = true;
  t.in[1] = xi;
  t.in[2] = yi;
  t.in[3] = xi && yi;
  const int xc = xi ? x1 : x0, yc = yi ? y1 : y0;
  const size_t base = (size_t)b * H * W;
  t.off[0] = (unsigned)((base + (size_t)y0 * W + x0) * C3);   // B H W 3C < 2^32 (checked by the entry point)
  t.off[1] = (unsigned)((base + (size_t)y0 * W + xc) * C3);
  t.off[2] = (unsigned)((base + (size_t)yc * W + x0) * C3);
  t.off[3] = (unsigned)((base + (size_t)yc * W + xc) * C3);
  return t;
}
}  // namespace

__global__ __launch_bounds__(kBlock) void ba_sample_stats_kernel(const float* __restrict__ conv1, const float* __restrict__ conv2,
                                                                 const float* __restrict__ px, const float* __restrict__ py,
                                                                 int N, int C, int H, int W, float* __restrict__ stats,
                                                                 float* __restrict__ absd_part, int G) {
  __shared__ float sAbs[kNumWaves][kMaxCJ * 64];
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, w = wave_id();
  const int CJ = (C + 63) >> 6, C3 = 3 * C;
  float absacc[kMaxCJ] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < kPixPerWave; ++i) {
    const int n = g * kPixPerBlock + w * kPixPerWave + i;
    if (n >= N) break;                                   // wave-uniform
    const size_t q = (size_t)b * N + n;
    const Taps t = make_taps(px[q], py[q], b, H, W, C3);
    float m11 = 0.f, m12 = 0.f, m22 = 0.f, g1 = 0.f, g2 = 0.f;
    if (t.m) {
      for (int j = 0; j < CJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
          float f2 = 0.f, gx = 0.f, gy = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float* r = conv2 + (size_t)t.off[k];
            const float wk = t.in[k] ? t.w[k] : 0.f;
            f2 = fmaf(wk, r[c], f2);
            gx = fmaf(wk, r[C + c], gx);
            gy = fmaf(wk, r[2 * C + c], gy);
          }
          const float d = conv1[q * C + c] - f2;         // bundlenet.py:234
          m11 = fmaf(gx, gx, m11);
          m12 = fmaf(gx, gy, m12);
          m22 = fmaf(gy, gy, m22);
          g1 = fmaf(gx, d, g1);
          g2 = fmaf(gy, d, g2);
          absacc[j] += fabsf(d);
        }
      }
      m11 = wave_sum(m11);
      m12 = wave_sum(m12);
      m22 = wave_sum(m22);
      g1 = wave_sum(g1);
      g2 = wave_sum(g2);
    }
    if (lane == 0) {
      float4* o = reinterpret_cast<float4*>(stats + q * 8);
      o[0] = make_float4(m11, m12, m22, g1);
      o[1] = make_float4(g2, t.m ? 1.f : 0.f, 0.f, 0.f);
    }
  }
  for (int j = 0; j < kMaxCJ; ++j) sAbs[w][lane + 64 * j] = absacc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kBlock)          // the block's 4 waves in fixed order
    absd_part[((size_t)b * G + g) * C + c] = ((sAbs[0][c] + sAbs[1][c]) + sAbs[2][c]) + sAbs[3][c];
}

__global__ __launch_bounds__(kBlock) void ba_sample_stats_grad_kernel(const float* __restrict__ conv1, const float* __restrict__ conv2,
                                                                      const float* __restrict__ px, const float* __restrict__ py,
                                                                      int N, int C, int H, int W, const float* __restrict__ dstats,
                                                                      const float* __restrict__ dabs, float* __restrict__ dconv1,
                                                                      float* __restrict__ dconv2, float* __restrict__ dpos) {
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, w = wave_id();
  const int CJ = (C + 63) >> 6, C3 = 3 * C;
  float da[kMaxCJ];
  for (int j = 0; j < kMaxCJ; ++j) {
    const int c = lane + 64 * j;
    da[j] = c < C ? dabs[(size_t)b * C + c] : 0.f;
  }
  for (int i = 0; i < kPixPerWave; ++i) {
    const int n = g * kPixPerBlock + w * kPixPerWave + i;
    if (n >= N) break;
    const size_t q = (size_t)b * N + n;
    const Taps t = make_taps(px[q], py[q], b, H, W, C3);
    float dpx = 0.f, dpy = 0.f;
    if (!t.m) {
      for (int j = 0; j < CJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C) dconv1[q * C + c] = 0.f;
      }
    } else {
      const float4 s0 = *reinterpret_cast<const float4*>(dstats + q * 8);
      const float dm11 = s0.x, dm12 = s0.y, dm22 = s0.z, dg1 = s0.w, dg2 = dstats[q * 8 + 4];
      for (int j = 0; j < CJ; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
          float v[4][3];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float* r = conv2 + (size_t)t.off[k];
            v[k][0] = t.in[k] ? r[c] : 0.f;
            v[k][1] = t.in[k] ? r[C + c] : 0.f;
            v[k][2] = t.in[k] ? r[2 * C + c] : 0.f;
          }
          float s[3];
#pragma unroll
          for (int e = 0; e < 3; ++e) s[e] = t.w[0] * v[0][e] + t.w[1] * v[1][e] + t.w[2] * v[2][e] + t.w[3] * v[3][e];
          const float gx = s[1], gy = s[2], d = conv1[q * C + c] - s[0];
          const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
          const float dd = gx * dg1 + gy * dg2 + sgn * da[j];
          float dv[3];
          dv[0] = -dd;                                              // d = conv1 - f2
          dv[1] = 2.f * gx * dm11 + gy * dm12 + d * dg1;
          dv[2] = 2.f * gy * dm22 + gx * dm12 + d * dg2;
          dconv1[q * C + c] = dd;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (t.in[k] && t.w[k] != 0.f) {
              float* r = dconv2 + (size_t)t.off[k];
              atomicAdd(r + c, t.w[k] * dv[0]);
              atomicAdd(r + C + c, t.w[k] * dv[1]);
              atomicAdd(r + 2 * C + c, t.w[k] * dv[2]);
            }
          }
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            dpx = fmaf(dv[e], (1.f - t.ay) * (v[1][e] - v[0][e]) + t.ay * (v[3][e] - v[2][e]), dpx);
            dpy = fmaf(dv[e], (1.f - t.ax) * (v[2][e] - v[0][e]) + t.ax * (v[3][e] - v[1][e]), dpy);
          }
        }
      }
      dpx = wave_sum(dpx);
      dpy = wave_sum(dpy);
    }
    if (lane == 0) *reinterpret_cast<float2*>(dpos + q * 2) = make_float2(dpx, dpy);
  }
}

int sample_stats_blocks(int N) { return (N + kPixPerBlock - 1) / kPixPerBlock; }

int launch_sample_stats(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C, int H,
                        int W, float* stats, float* absd_part, hipStream_t s) {
  if (C < 1 || C > 64 * kMaxCJ) return BANET_ERR_UNSUPPORTED;
  const int G = sample_stats_blocks(N);
  hipLaunchKernelGGL(ba_sample_stats_kernel, dim3(G, B), dim3(kBlock), 0, s, conv1, conv2, px, py, N, C, H, W, stats, absd_part, G);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

int launch_sample_stats_grad(const float* conv1, const float* conv2, const float* px, const float* py, int B, int N, int C,
                             int H, int W, const float* dstats, const float* dabs, float* dconv1, float* dconv2, float* dpos,
                             hipStream_t s) {
  if (C < 1 || C > 64 * kMaxCJ) return BANET_ERR_UNSUPPORTED;
  const int G = sample_stats_blocks(N);
  hipLaunchKernelGGL(ba_sample_stats_grad_kernel, dim3(G, B), dim3(kBlock), 0, s, conv1, conv2, px, py, N, C, H, W, dstats, dabs,
                     dconv1, dconv2, dpos);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
