// Per-level preparation kernels -- the step immediately before the LM loop:
//   ba_resample_kernel     tf.contrib.resampler.resampler (bundlenet.py:290,320,343-344,385) and
//                          interpolate2d2 (legacy/utils_python.py:177-232; legacy/ba.py:115)
//   ba_target_map_kernel   conv2 = [f | gx | gy], grad_fixed with REFLECT padding
//                          (bundlenet.py:92-100,323-324 ; legacy/ba.py:17-25,116-118)
//   ba_depth_output_kernel depth = init + basis . W   (bundlenet.py:397)
// All three are pure streaming kernels (HBM-bound): channel-contiguous rows, lane = channel(s),
// so every tap / row is a coalesced load.  Same operation order as the reference expressions
// (the oracle restates them), fp32.
#include "kernels.hpp"

namespace banet {

// mode 0: tf.contrib.resampler -- zero padding, a point is sampled iff x > -1, y > -1, x < W, y < H;
//         weights from the CEIL side (dx = cx - x), sum order a*f(fx,fy) + b*f(cx,cy) + c*f(fx,cy) + d*f(cx,fy)
// mode 1: interpolate2d2 -- weights from the unclamped floor, indices clamped, ((a+b)+c)+d
__global__ __launch_bounds__(256) void ba_resample_kernel(const float* __restrict__ data, const float* __restrict__ warp,
                                                          float* __restrict__ out, int N, int C, int H, int W, int mode) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  const float* __restrict__ img = data + (size_t)b * H * W * C;
  for (int n = wv; n < N; n += nw) {   // one wave per point, lanes over channels
    const float x = warp[((size_t)b * N + n) * 2], y = warp[((size_t)b * N + n) * 2 + 1];
    float* o = out + ((size_t)b * N + n) * C;
    if (mode == 0) {
      const bool ok = (x > -1.f) && (y > -1.f) && (x < (float)W) && (y < (float)H);
      const float xs = ok ? x : 0.f, ys = ok ? y : 0.f;
      const float fxf = floorf(xs), fyf = floorf(ys), cxf = fxf + 1.f, cyf = fyf + 1.f;
      const float dx = cxf - xs, dy = cyf - ys;
      const int fx = (int)fxf, fy = (int)fyf, cx = (int)cxf, cy = (int)cyf;
      auto in = [&](int xi, int yi) { return xi >= 0 && yi >= 0 && xi <= W - 1 && yi <= H - 1; };
      auto at = [&](int xi, int yi) { return img + ((size_t)min(max(yi, 0), H - 1) * W + min(max(xi, 0), W - 1)) * C; };
      const float* p00 = at(fx, fy);
      const float* p11 = at(cx, cy);
      const float* p01 = at(fx, cy);
      const float* p10 = at(cx, fy);
      const float m00 = in(fx, fy) ? 1.f : 0.f, m11 = in(cx, cy) ? 1.f : 0.f, m01 = in(fx, cy) ? 1.f : 0.f,
                  m10 = in(cx, fy) ? 1.f : 0.f;
      const float wa = dx * dy, wb = (1.f - dx) * (1.f - dy), wc = dx * (1.f - dy), wd = (1.f - dx) * dy;
      for (int c = lane; c < C; c += 64) {
        const float v = ((wa * (m00 * p00[c]) + wb * (m11 * p11[c])) + wc * (m01 * p01[c])) + wd * (m10 * p10[c]);
        o[c] = ok ? v : 0.f;
      }
    } else {
      const float x0f = floorf(x), y0f = floorf(y);
      const float dx = x - x0f, dy = y - y0f;
      const float w00 = (1.f - dx) * (1.f - dy), w01 = dx * (1.f - dy), w10 = (1.f - dx) * dy, w11 = dx * dy;
      // NaN / inf coordinates: index 0 / saturated, like the oracle's nan_to_num before the clamp
      const float xc = (x0f == x0f) ? fminf(fmaxf(x0f, -1e9f), 1e9f) : 0.f, yc = (y0f == y0f) ? fminf(fmaxf(y0f, -1e9f), 1e9f) : 0.f;
      const int x0 = (int)xc, y0 = (int)yc;
      const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
      const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
      const float* p00 = img + ((size_t)ya * W + xa) * C;
      const float* p01 = img + ((size_t)ya * W + xb) * C;
      const float* p10 = img + ((size_t)yb * W + xa) * C;
      const float* p11 = img + ((size_t)yb * W + xb) * C;
      for (int c = lane; c < C; c += 64) o[c] = ((p00[c] * w00 + p01[c] * w01) + p10[c] * w10) + p11[c] * w11;
    }
  }
}

// [B,H,W,C] -> [B,H,W,3C] = [f | gx | gy]; thread = (pixel, channel), channels fastest
__global__ __launch_bounds__(256) void ba_target_map_kernel(const float* __restrict__ img, float* __restrict__ out, int H,
                                                            int W, int C) {
  const int b = blockIdx.y;
  const size_t total = (size_t)H * W * C;
  const float* __restrict__ im = img + (size_t)b * total;
  float* __restrict__ o = out + (size_t)b * total * 3;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const size_t pix = e / C;
    const int x = (int)(pix % W), y = (int)(pix / W);
    const float f = im[e];
    const float gx = 0.5f * (im[((size_t)y * W + refl_p(x, W)) * C + c] - im[((size_t)y * W + refl_m(x)) * C + c]);
    const float gy = 0.5f * (im[((size_t)refl_p(y, H) * W + x) * C + c] - im[((size_t)refl_m(y) * W + x) * C + c]);
    float* op = o + pix * 3 * C;
    op[c] = f;
    op[C + c] = gx;
    op[2 * C + c] = gy;
  }
}

// out[b,n] = init[b,n] + sum_k basis[b,n,k] W[b,k]: one wave per row, fixed-order lane reduction
__global__ __launch_bounds__(256) void ba_depth_output_kernel(const float* __restrict__ init, const float* __restrict__ basis,
                                                              const float* __restrict__ Wc, float* __restrict__ out, int N,
                                                              int K) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  const float* __restrict__ wb = Wc + (size_t)b * K;
  for (int n = wv; n < N; n += nw) {
    const float* row = basis + ((size_t)b * N + n) * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(row[k], wb[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[(size_t)b * N + n] = init[(size_t)b * N + n] + acc;
  }
}

int launch_resample(const float* data, const float* warp, float* out, int B, int N, int C, int H, int W, int mode,
                    hipStream_t s) {
  if (B <= 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return BANET_ERR_INVALID_ARG;
  int gx = (N + 3) / 4;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(ba_resample_kernel, dim3(gx, B), dim3(256), 0, s, data, warp, out, N, C, H, W, mode);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

int launch_target_map(const float* img, float* out, int B, int H, int W, int C, hipStream_t s) {
  if (B <= 0 || C <= 0 || H < 2 || W < 2) return BANET_ERR_INVALID_ARG;
  size_t blocks = ((size_t)H * W * C + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(ba_target_map_kernel, dim3((unsigned)blocks, B), dim3(256), 0, s, img, out, H, W, C);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

int launch_depth_output(const float* init, const float* basis, const float* Wc, float* out, int B, int N, int K,
                        hipStream_t s) {
  if (B <= 0 || N <= 0 || K <= 0) return BANET_ERR_INVALID_ARG;
  int gx = (N + 3) / 4;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(ba_depth_output_kernel, dim3(gx, B), dim3(256), 0, s, init, basis, Wc, out, N, K);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
