// lambda prediction, damping, linear solve and SE(3)/W update -- one workgroup per window.
//
// Restates (citations under /root/reference):
//   avg residual & lambda MLP    bundlenet.py:165-173,241-253 ; legacy/ba.py:187-190,266-275
//   damping                      bundlenet.py:181-182,264-266 ; legacy/ba.py:200-201,285-286
//   solve                        tf.matrix_solve (LU, partial pivoting) bundlenet.py:183,267 ;
//                                tf.qr + triangular solve legacy/ba.py:292-293
//   update                       bundlenet.py:185-190,269-276 ; legacy/ba.py:208-213,295-302
//   accept / terminate           legacy/ba.py:132-140,304-345
// The matrix lives in LDS (P <= 134 -> 72 KB of the 160 KB per CU).
#include "kernels.hpp"

namespace banet {

constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;
constexpr float kSeluScale = 1.0507009873554804934193349852946f;
constexpr float kAngleChange = (float)(0.002 * (3.14 / 180.0));  // legacy/ba.py:6
constexpr float kTranslationChange = 0.0002f;                     // legacy/ba.py:7
constexpr float kResidualRatio = 1.0f;                            // legacy/ba.py:8

__device__ __forceinline__ float selu(float x) {
  return kSeluScale * (x > 0.f ? x : kSeluAlpha * (expf(x) - 1.f));
}

// block-wide sum of one value per thread, fixed order (deterministic)
__device__ float block_sum(float v, float* sred) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  return r;
}

// one k=1 conv layer: out[o] = act(sum_i in[i] W[i][o] + b[o]);  act: 0 selu, 1 tanh
__device__ void mlp_layer(const float* in, float* out, const float* __restrict__ Wt, const float* __restrict__ bias,
                          int nin, int nout, int act, float* sred) {
  const int tid = threadIdx.x;
  if (nout >= 64) {
    for (int o = tid; o < nout; o += kBlock) {
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
      for (int i = tid; i < nin; i += kBlock) a = fmaf(in[i], Wt[(size_t)i * nout + o], a);
      const float s = block_sum(a, sred);
      if (tid == 0) {
        const float v = s + bias[o];
        out[o] = act == 0 ? selu(v) : tanhf(v);
      }
    }
    __syncthreads();
  }
}

// Householder QR of a 6x6 system and back-substitution (thread 0, LDS-resident, fp32):
// the algorithm class of tf.qr + tf.linalg.solve on the triangular factor.
__device__ void qr_solve_small(float* A, int ld, float* rhs, int n, float* x) {
  for (int k = 0; k < n; ++k) {
    float nrm = 0.f;
    for (int i = k; i < n; ++i) nrm += A[i * ld + k] * A[i * ld + k];
    nrm = sqrtf(nrm);
    if (nrm == 0.f) continue;
    const float akk = A[k * ld + k];
    const float alpha = akk > 0.f ? -nrm : nrm;
    // v = a_k - alpha e_k  (stored over the column), beta = 2 / v^T v
    A[k * ld + k] = akk - alpha;
    float vtv = 0.f;
    for (int i = k; i < n; ++i) vtv += A[i * ld + k] * A[i * ld + k];
    const float beta = 2.f / vtv;
    for (int j = k + 1; j < n; ++j) {
      float s = 0.f;
      for (int i = k; i < n; ++i) s += A[i * ld + k] * A[i * ld + j];
      s *= beta;
      for (int i = k; i < n; ++i) A[i * ld + j] -= s * A[i * ld + k];
    }
    float s = 0.f;
    for (int i = k; i < n; ++i) s += A[i * ld + k] * rhs[i];
    s *= beta;
    for (int i = k; i < n; ++i) rhs[i] -= s * A[i * ld + k];
    A[k * ld + k] = alpha;  // R's diagonal
  }
  for (int k = n - 1; k >= 0; --k) {
    float s = rhs[k];
    for (int j = k + 1; j < n; ++j) s -= A[k * ld + j] * x[j];
    x[k] = s / A[k * ld + k];
  }
}

// LU with partial pivoting on the augmented matrix [A | b] in LDS, all 256 threads.
__device__ void lu_solve(float* A, int ld, int n, float* x, int* spiv) {
  const int tid = threadIdx.x;
  for (int k = 0; k < n; ++k) {
    // pivot search in column k (wave 0)
    if (tid < 64) {
      float best = -1.f;
      int bi = k;
      for (int i = k + tid; i < n; i += 64) {
        const float v = fabsf(A[i * ld + k]);
        if (v > best) {
          best = v;
          bi = i;
        }
      }
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) {
        const float ob = __shfl_xor(best, s, 64);
        const int oi = __shfl_xor(bi, s, 64);
        if (ob > best || (ob == best && oi < bi)) {
          best = ob;
          bi = oi;
        }
      }
      if (tid == 0) *spiv = bi;
    }
    __syncthreads();
    const int piv = *spiv;
    if (piv != k) {
      for (int j = tid; j <= n; j += kBlock) {
        const float t0 = A[k * ld + j];
        A[k * ld + j] = A[piv * ld + j];
        A[piv * ld + j] = t0;
      }
    }
    __syncthreads();
    const float inv = 1.f / A[k * ld + k];
    __syncthreads();
    for (int i = k + 1 + tid; i < n; i += kBlock) A[i * ld + k] *= inv;
    __syncthreads();
    const int m = n - k - 1;       // rows below
    const int cols = n - k;        // columns k+1..n (n = augmented rhs)
    for (int e = tid; e < m * cols; e += kBlock) {
      const int i = k + 1 + e / cols, j = k + 1 + (e - (e / cols) * cols);
      A[i * ld + j] = fmaf(-A[i * ld + k], A[k * ld + j], A[i * ld + j]);
    }
    __syncthreads();
  }
  // back substitution, column oriented
  for (int k = n - 1; k >= 0; --k) {
    if (tid == 0) x[k] = A[k * ld + n] / A[k * ld + k];
    __syncthreads();
    const float xk = x[k];
    for (int i = tid; i < k; i += kBlock) A[i * ld + n] = fmaf(-A[i * ld + k], xk, A[i * ld + n]);
    __syncthreads();
  }
}

__device__ void rodrigues(const float w[3], bool clamp, float Rw[9], float V[9]) {
  // exp(w): bundlenet.py:17-37 / legacy/ba.py:60-80 ; V(w): bundlenet.py:39-46
  const float th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  float th = sqrtf(th2);
  const float thv = th;  // VMatrix uses the unclamped angle
  if (clamp) th = fmaxf(th, 1e-6f);
  if (!(th > 0.f)) {  // zero update: the reference divides 0/0 here; return identity
    for (int i = 0; i < 9; ++i) Rw[i] = V[i] = (i % 4 == 0) ? 1.f : 0.f;
    return;
  }
  const float kx = w[0] / th, ky = w[1] / th, kz = w[2] / th;
  const float c = cosf(th), s = sinf(th), oc = 1.f - c;
  Rw[0] = c + kx * kx * oc;
  Rw[1] = kx * ky * oc - kz * s;
  Rw[2] = ky * s + kx * kz * oc;
  Rw[3] = kz * s + kx * ky * oc;
  Rw[4] = c + ky * ky * oc;
  Rw[5] = -kx * s + ky * kz * oc;
  Rw[6] = -ky * s + kx * kz * oc;
  Rw[7] = kx * s + ky * kz * oc;
  Rw[8] = c + kz * kz * oc;
  if (!(thv > 0.f)) {
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.f : 0.f;
    return;
  }
  const float cv = cosf(thv), sv = sinf(thv);
  const float a = (1.f - cv) / (thv * thv), bq = (thv - sv) / (thv * thv * thv);
  const float Kx[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
  float K2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K2[i * 3 + j] = Kx[i * 3] * Kx[j] + Kx[i * 3 + 1] * Kx[3 + j] + Kx[i * 3 + 2] * Kx[6 + j];
  for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + a * Kx[i] + bq * K2[i];
}

__global__ __launch_bounds__(kBlock) void ba_solve_update_kernel(const SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int P = a.P, C = a.C, K = a.K, ld = P + 2;
  float* sA = smem;                 // [P][ld] augmented
  float* sX = sA + P * ld;          // [P]
  float* sH0 = sX + ((P + 3) & ~3); // MLP ping
  float* sH1 = sH0 + 4 * C;         // MLP pong
  float* sAvg = sH1 + 4 * C;        // [C]
  float* sRed = sAvg + C;           // [8]
  int* sPiv = reinterpret_cast<int*>(sRed + 8);
  float* sScal = sRed + 12;         // lambda, avg_scalar, flags

  LmCtl* ctl = a.ctl ? a.ctl + b : nullptr;
  if (ctl && ctl->active == 0) return;

  const bool legacy = a.variant == BANET_LEGACY_LM || a.variant == BANET_LEGACY_FIXED;
  const float Nf = (float)a.N;
  const float nval = a.nvalid[b];
  // ---- average residual -------------------------------------------------------------
  const float numvalid = Nf / nval;  // legacy/ba.py:257
  float ss = 0.f, sm = 0.f;
  for (int c = tid; c < C; c += kBlock) {
    float v = a.absres[(size_t)b * C + c] / Nf;                 // reduce_mean over N
    if (a.variant == BANET_LEGACY_LM) v = numvalid * v;        // legacy/ba.py:268
    sAvg[c] = v;
    ss += v * v;
    sm += v;
  }
  const float nrm = sqrtf(block_sum(ss, sRed));
  const float avg_scalar = block_sum(sm, sRed) / (float)C;     // legacy/ba.py:275
  // ---- lambda -----------------------------------------------------------------------
  float lam;
  if (a.use_mlp) {
    __syncthreads();
    mlp_layer(sAvg, sH0, a.mlp.w[0], a.mlp.b[0], C, 2 * C, 0, sRed);
    mlp_layer(sH0, sH1, a.mlp.w[1], a.mlp.b[1], 2 * C, 4 * C, 0, sRed);
    mlp_layer(sH1, sH0, a.mlp.w[2], a.mlp.b[2], 4 * C, 2 * C, 0, sRed);
    mlp_layer(sH0, sH1, a.mlp.w[3], a.mlp.b[3], 2 * C, C, 0, sRed);
    mlp_layer(sH1, sH0, a.mlp.w[4], a.mlp.b[4], C, 1, 1, sRed);
    const float y = sH0[0];
    const float e = legacy ? 1.f : 2.f;                          // ba.py:274 / bundlenet.py:173,249
    lam = powf(nrm, e + y);
  } else {
    lam = powf(nrm, 2.f);                                        // legacy/ba.py:190
  }
  if (a.variant == BANET_BUNDLE) lam *= a.l2_base;               // bundlenet.py:252-253
  // ---- accept / terminate (legacy early termination) ----------------------------------
  if (ctl) {
    if (tid == 0) {
      int go = 1;
      if (ctl->pending) {
        if (!(avg_scalar < kResidualRatio * ctl->avg_prev)) {  // reject: legacy/ba.py:343-345
          for (int i = 0; i < 9; ++i) a.st.R[b * 9 + i] = ctl->Rprev[i];
          for (int i = 0; i < 3; ++i) a.st.T[b * 3 + i] = ctl->Tprev[i];
          ctl->uw = 0.f;
          ctl->ut = 0.f;
        }
        ctl->pending = 0;
      }
      // loop condition, legacy/ba.py:132-133
      if (!(a.st.iters[b] < a.max_iters && kAngleChange < ctl->uw && kTranslationChange < ctl->ut)) {
        ctl->active = 0;
        go = 0;
      }
      sScal[2] = (float)go;
    }
    __syncthreads();
    if (sScal[2] == 0.f) return;
  }
  // ---- damping ------------------------------------------------------------------------
  const float* A_g = a.AtA + (size_t)b * P * P;
  for (int e = tid; e < P * P; e += kBlock) {
    const int i = e / P, j = e - i * P;
    float v = A_g[e];
    if (i == j && !(a.variant == BANET_BUNDLE && i == P - 1)) v = v + (v + 1e-5f) * lam;
    sA[i * ld + j] = v;
  }
  for (int i = tid; i < P; i += kBlock) sA[i * ld + P] = a.Atb[(size_t)b * P + i];
  __syncthreads();
  // ---- solve --------------------------------------------------------------------------
  if (legacy && P == 6) {
    if (tid == 0) {
      float rhs[6];
      for (int i = 0; i < 6; ++i) rhs[i] = sA[i * ld + P];
      qr_solve_small(sA, ld, rhs, 6, sX);
    }
    __syncthreads();
  } else {
    lu_solve(sA, ld, P, sX, sPiv);
  }
  // ---- update -------------------------------------------------------------------------
  for (int k = tid; k < P; k += kBlock) a.st.delta[(size_t)b * P + k] = sX[k];
  for (int k = tid; k < K; k += kBlock) a.st.Wc[(size_t)b * K + k] += sX[6 + k];   // bundlenet.py:276
  if (tid == 0) {
    float w[3] = {sX[0], sX[1], sX[2]}, t[3] = {sX[3], sX[4], sX[5]};
    float Rw[9], V[9], Ro[9], To[3], Rn[9], Tn[3];
    for (int i = 0; i < 9; ++i) Ro[i] = a.st.R[b * 9 + i];
    for (int i = 0; i < 3; ++i) To[i] = a.st.T[b * 3 + i];
    rodrigues(w, !legacy, Rw, V);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = Rw[i * 3] * Ro[j] + Rw[i * 3 + 1] * Ro[3 + j] + Rw[i * 3 + 2] * Ro[6 + j];
    for (int i = 0; i < 3; ++i) {
      const float rt = Rw[i * 3] * To[0] + Rw[i * 3 + 1] * To[1] + Rw[i * 3 + 2] * To[2];
      const float vt = (a.variant == BANET_LEGACY_FIXED)
                           ? t[i]                                                     // legacy/ba.py:213
                           : V[i * 3] * t[0] + V[i * 3 + 1] * t[1] + V[i * 3 + 2] * t[2];
      Tn[i] = vt + rt;
    }
    if (ctl) {
      for (int i = 0; i < 9; ++i) ctl->Rprev[i] = Ro[i];
      for (int i = 0; i < 3; ++i) ctl->Tprev[i] = To[i];
      ctl->avg_prev = avg_scalar;
      ctl->uw = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      ctl->ut = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
      ctl->pending = 1;
    }
    for (int i = 0; i < 9; ++i) a.st.R[b * 9 + i] = Rn[i];
    for (int i = 0; i < 3; ++i) a.st.T[b * 3 + i] = Tn[i];
    a.st.iters[b] += 1;
    a.st.ratio[b] = (a.variant == BANET_LEGACY_FIXED) ? nval / Nf : numvalid;   // ba.py:214 / :344
    a.st.lambda_out[b] = lam;
  }
}

__global__ void lm_ctl_init_kernel(LmCtl* ctl, int32_t* iters, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  ctl[b].active = 1;
  ctl[b].pending = 0;
  ctl[b].avg_prev = 0.f;
  ctl[b].uw = 1.f;   // legacy/ba.py:128-129
  ctl[b].ut = 1.f;
  iters[b] = 0;
}

__global__ void zero_iters_kernel(int32_t* iters, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) iters[b] = 0;
}

size_t solve_lds_bytes(int P, int C) {
  const size_t fl = (size_t)P * (P + 2) + ((P + 3) & ~3) + 8 * C + C + 8 + 8;
  return fl * sizeof(float);
}

int launch_solve(const SolveArgs& a, hipStream_t s) {
  const size_t lds = solve_lds_bytes(a.P, a.C);
  if (lds > 160 * 1024) return BANET_ERR_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)ba_solve_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ba_solve_update_kernel, dim3(a.B), dim3(kBlock), lds, s, a);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

void launch_ctl_init(LmCtl* ctl, int32_t* iters, int B, hipStream_t s) {
  hipLaunchKernelGGL(lm_ctl_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ctl, iters, B);
}

void launch_zero_iters(int32_t* iters, int B, hipStream_t s) {
  hipLaunchKernelGGL(zero_iters_kernel, dim3((B + 63) / 64), dim3(64), 0, s, iters, B);
}

}  // namespace banet
