// The lambda predictor's building blocks (bundlenet.py:168-172: five k = 1 conv layers, selu x4, tanh), templated on the
// workgroup size: ba_solve_update_kernel evaluates it with 1024 threads, the MLP role workgroup of the SYRK launch with 256.
#pragma once
#include "kernels.hpp"

namespace banet {

constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;
constexpr float kSeluScale = 1.0507009873554804934193349852946f;
// (the loop thresholds / residual ratio / solver choice of legacy/ba.py:5-9 arrive in SolveArgs::lm)

__device__ __forceinline__ float selu(float x) {
  return kSeluScale * (x > 0.f ? x : kSeluAlpha * (expf(x) - 1.f));
}

// block-wide sum of one value per thread, fixed order (deterministic)
template <int NT>
__device__ float block_sum_t(float v, float* sred) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += sred[i];
  return r;
}

// one k=1 conv layer: out[o] = act(sum_i in[i] W[i][o] + b[o]);  act: 0 selu, 1 tanh.
// Output quads x input slices over the 256 threads: 16-byte weight loads, 8 in flight per thread,
// partial sums combined in fixed order through LDS (sPart: >= 1024 floats).
template <int NT>
__device__ void mlp_layer_t(const float* in, float* out, const float* __restrict__ Wt, const float* __restrict__ bias,
                          int nin, int nout, int act, float* sPart, float* sred) {
  const int tid = threadIdx.x;
  const int groups = nout >> 2;
  if ((nout & 3) == 0 && groups <= NT && ((reinterpret_cast<uintptr_t>(Wt) & 15) == 0)) {
    int ksplit = NT / groups;
    if (ksplit > nin) ksplit = nin;
    const int q = tid % groups, sl = tid / groups;
    if (sl < ksplit) {
      const int chunk = (nin + ksplit - 1) / ksplit;
      const int i0 = sl * chunk, i1 = min(nin, i0 + chunk);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* wp = Wt + 4 * q;
#pragma unroll 16
      for (int i = i0; i < i1; ++i) {
        const float4 w4 = *reinterpret_cast<const float4*>(wp + (size_t)i * nout);
        const float xv = in[i];
        acc.x = fmaf(xv, w4.x, acc.x);
        acc.y = fmaf(xv, w4.y, acc.y);
        acc.z = fmaf(xv, w4.z, acc.z);
        acc.w = fmaf(xv, w4.w, acc.w);
      }
      *reinterpret_cast<float4*>(sPart + sl * nout + 4 * q) = acc;
    }
    __syncthreads();
    for (int o = tid; o < nout; o += NT) {
      float v = 0.f;
      for (int k = 0; k < ksplit; ++k) v += sPart[k * nout + o];
      v += bias[o];
      out[o] = act == 0 ? selu(v) : tanhf(v);
    }
    __syncthreads();
  } else if (nout >= 64) {
    for (int o = tid; o < nout; o += NT) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int i = 0;
      for (; i + 3 < nin; i += 4) {
        a0 = fmaf(in[i], Wt[(size_t)i * nout + o], a0);
        a1 = fmaf(in[i + 1], Wt[(size_t)(i + 1) * nout + o], a1);
        a2 = fmaf(in[i + 2], Wt[(size_t)(i + 2) * nout + o], a2);
        a3 = fmaf(in[i + 3], Wt[(size_t)(i + 3) * nout + o], a3);
      }
      for (; i < nin; ++i) a0 = fmaf(in[i], Wt[(size_t)i * nout + o], a0);
      const float v = ((a0 + a1) + (a2 + a3)) + bias[o];
      out[o] = act == 0 ? selu(v) : tanhf(v);
    }
    __syncthreads();
  } else {
    for (int o = 0; o < nout; ++o) {
      float a = 0.f;
      for (int i = tid; i < nin; i += NT) a = fmaf(in[i], Wt[(size_t)i * nout + o], a);
      const float s = block_sum_t<NT>(a, sred);
      if (tid == 0) {
        const float v = s + bias[o];
        out[o] = act == 0 ? selu(v) : tanhf(v);
      }
    }
    __syncthreads();
  }
}


// MLP role workgroup (256 threads) of the SYRK launch: y = MLP(avg) for window b from the folded gather partials, so that the
// solve kernel finds it ready instead of streaming the 1.3 MB of weights itself (19 us of its 79).  avg_c = sum_n |d_nc| / N with
// the sum taken over the partial rows in ba_reduce2_kernel's order (bit-identical to the absres the solve kernel reads).
constexpr int kMlpRoleFloats = 256 + 1024 + 1024 + 4096 + 32;
__device__ inline void mlp_role_block(const MlpRole& r, int b, float* sm) {
  const int tid = threadIdx.x, C = r.C;
  float* sAvg = sm;            // [C <= 256]
  float* sH0 = sAvg + 256;     // [4 C]
  float* sH1 = sH0 + 1024;     // [4 C]
  float* sPart = sH1 + 1024;   // [4096]
  float* sRed = sPart + 4096;  // [32]
  for (int c = tid; c < C; c += 256) {
    float v = 0.f;
    for (int pp = 0; pp < r.pairs; ++pp) {
      const float* p = r.gpart + ((size_t)b * r.pairs + pp) * r.grows * r.gstride + kGHdr + c;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int i = 0;
      for (; i + 3 < r.grows; i += 4) {
        s0 += p[(size_t)(i + 0) * r.gstride];
        s1 += p[(size_t)(i + 1) * r.gstride];
        s2 += p[(size_t)(i + 2) * r.gstride];
        s3 += p[(size_t)(i + 3) * r.gstride];
      }
      for (; i < r.grows; ++i) s0 += p[(size_t)i * r.gstride];
      v += (s0 + s1) + (s2 + s3);
    }
    sAvg[c] = v / r.Nf;
  }
  __syncthreads();
  mlp_layer_t<256>(sAvg, sH0, r.mlp.w[0], r.mlp.b[0], C, 2 * C, 0, sPart, sRed);
  mlp_layer_t<256>(sH0, sH1, r.mlp.w[1], r.mlp.b[1], 2 * C, 4 * C, 0, sPart, sRed);
  mlp_layer_t<256>(sH1, sH0, r.mlp.w[2], r.mlp.b[2], 4 * C, 2 * C, 0, sPart, sRed);
  mlp_layer_t<256>(sH0, sH1, r.mlp.w[3], r.mlp.b[3], 2 * C, C, 0, sPart, sRed);
  mlp_layer_t<256>(sH1, sH0, r.mlp.w[4], r.mlp.b[4], C, 1, 1, sPart, sRed);
  if (tid == 0) r.y[b] = sH0[0];
}

}  // namespace banet
