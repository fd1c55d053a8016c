// lambda prediction, damping, linear solve and SE(3)/W update -- one workgroup per window.
//
// Restates (citations under /root/reference):
//   avg residual & lambda MLP    bundlenet.py:165-173,241-253 ; legacy/ba.py:187-190,266-275
//   damping                      bundlenet.py:181-182,264-266 ; legacy/ba.py:200-201,285-286
//   solve                        tf.matrix_solve (LU, partial pivoting) bundlenet.py:183,267 ;
//                                tf.qr + triangular solve legacy/ba.py:292-293
//   update                       bundlenet.py:185-190,269-276 ; legacy/ba.py:208-213,295-302
//   accept / terminate           legacy/ba.py:132-140,304-345
// The matrix lives in LDS (P <= 134 -> 72 KB of the 160 KB per CU); larger systems (K = 256, many frames) keep it in
// the caller's workspace (template<bool BIG>, workgroup-private and L2-resident).
#include <type_traits>

#include "kernels.hpp"
#include "mlp.hpp"

namespace banet {

#ifdef BANET_TIMING
__device__ __forceinline__ unsigned long long stick() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define STICK(v) const unsigned long long v = stick()
#else
#define STICK(v)
#endif

constexpr int kSolveThreads = 1024;   // 16 waves: 4 per SIMD, so the latency-bound phases overlap
[[maybe_unused]] constexpr int kSolveWaves = kSolveThreads / 64;
constexpr int kGrid = 32;              // 2-D cyclic thread grid of the register-resident solvers

// selu / block_sum_t / mlp_layer_t: mlp.hpp (shared with the MLP role workgroup of the SYRK launch)
__device__ __forceinline__ float block_sum(float v, float* sred) { return block_sum_t<kSolveThreads>(v, sred); }

// sum over the 64 lanes without the LDS crossbar; every lane gets the total
__device__ __forceinline__ float wave_sum_fast(float v) {
  v = row16_sum(v);
  v = bfly_merge(v, v, 16);
  return bfly_merge(v, v, 32);
}

// max / min over the 64 lanes on DPP + permlane swaps (every lane gets the result)
__device__ __forceinline__ float wave_max_fast(float v) {
  v = fmaxf(v, dpp_mov<kDppRor8>(v));
  v = fmaxf(v, dpp_mov<kDppHalfMirror>(v));
  v = fmaxf(v, dpp_mov<kDppXor2>(v));
  v = fmaxf(v, dpp_mov<kDppXor1>(v));
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = fmaxf(a, b);
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ int wave_min_fast(int v) {
  auto dpp = [](int x, auto ctrl) { return __builtin_amdgcn_update_dpp(0x7fffffff, x, decltype(ctrl)::value, 0xF, 0xF, false); };
  v = min(v, dpp(v, std::integral_constant<int, kDppRor8>{}));
  v = min(v, dpp(v, std::integral_constant<int, kDppHalfMirror>{}));
  v = min(v, dpp(v, std::integral_constant<int, kDppXor2>{}));
  v = min(v, dpp(v, std::integral_constant<int, kDppXor1>{}));
  int a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = min(a, b);
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return min(a, b);
}

__device__ __forceinline__ void mlp_layer(const float* in, float* out, const float* __restrict__ Wt, const float* __restrict__ bias,
                                          int nin, int nout, int act, float* sPart, float* sred) {
  mlp_layer_t<kSolveThreads>(in, out, Wt, bias, nin, nout, act, sPart, sred);
}

// Householder QR of the 6x6 legacy system and back-substitution (thread 0, fp32): the algorithm class of tf.qr +
// tf.linalg.solve on the triangular factor.  Round 6: the matrix lives in REGISTERS (every loop has compile-time bounds and is
// unrolled) -- the LDS-resident loop form cost ~400 dependent LDS round trips and, with inverse_solve_small's dynamically indexed
// rows, 644 scratch instructions in the kernel (round-5 review): same operations in the same order, so the same bits.
__device__ void qr_solve_small(const float* A_lds, int ld, const float* rhs_in, int /*n = 6*/, float* x) {
  constexpr int n = 6;
  float A[n][n], rhs[n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    rhs[i] = rhs_in[i];
#pragma unroll
    for (int j = 0; j < n; ++j) A[i][j] = A_lds[i * ld + j];
  }
#pragma unroll
  for (int k = 0; k < n; ++k) {
    float nrm = 0.f;
#pragma unroll
    for (int i = k; i < n; ++i) nrm += A[i][k] * A[i][k];
    nrm = sqrtf(nrm);
    if (nrm != 0.f) {
      const float akk = A[k][k];
      const float alpha = akk > 0.f ? -nrm : nrm;
      // v = a_k - alpha e_k  (stored over the column), beta = 2 / v^T v
      A[k][k] = akk - alpha;
      float vtv = 0.f;
#pragma unroll
      for (int i = k; i < n; ++i) vtv += A[i][k] * A[i][k];
      const float beta = 2.f / vtv;
#pragma unroll
      for (int j = k + 1; j < n; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = k; i < n; ++i) s += A[i][k] * A[i][j];
        s *= beta;
#pragma unroll
        for (int i = k; i < n; ++i) A[i][j] -= s * A[i][k];
      }
      float s = 0.f;
#pragma unroll
      for (int i = k; i < n; ++i) s += A[i][k] * rhs[i];
      s *= beta;
#pragma unroll
      for (int i = k; i < n; ++i) rhs[i] -= s * A[i][k];
      A[k][k] = alpha;  // R's diagonal
    }
  }
  float xs[n];
#pragma unroll
  for (int k = n - 1; k >= 0; --k) {
    float s = rhs[k];
#pragma unroll
    for (int j = k + 1; j < n; ++j) s -= A[k][j] * xs[j];
    xs[k] = s / A[k][k];
  }
#pragma unroll
  for (int k = 0; k < n; ++k) x[k] = xs[k];
}

// tf.matmul(tf.matrix_inverse(AtA), Atb) for the 6x6 legacy system (legacy/ba.py:203,290, `qr = False`): the explicit
// inverse by Gauss-Jordan elimination with partial pivoting (the algorithm class of tf.matrix_inverse: LU-PP solves
// against the identity), then the product.  Thread 0, registers: the pivot row is brought up by conditional swaps with
// compile-time row indices (the largest candidate ends in row k whatever the order of the others), so no row is addressed dynamically.
__device__ void inverse_solve_small(const float* A, int ld, const float* rhs, float* x) {
  float M[6][12];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      M[i][j] = A[i * ld + j];
      M[i][6 + j] = (i == j) ? 1.f : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const bool sw = fabsf(M[i][k]) > fabsf(M[k][k]);
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const float a = M[k][j], c = M[i][j];
        M[k][j] = sw ? c : a;
        M[i][j] = sw ? a : c;
      }
    }
    const float inv = 1.f / M[k][k];
#pragma unroll
    for (int j = 0; j < 12; ++j) M[k][j] *= inv;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i == k) continue;
      const float f = M[i][k];
#pragma unroll
      for (int j = 0; j < 12; ++j) M[i][j] = fmaf(-f, M[k][j], M[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) s = fmaf(M[i][6 + j], rhs[j], s);
    x[i] = s;
  }
}

// Register-resident LU with partial pivoting for n <= 16*NR - 1: the augmented matrix is
// distributed 2-D cyclically over the 16x16 thread grid (thread (ti,tj) owns rows ti+16r, columns
// tj+16c), so the trailing update runs out of registers.  Per elimination step: owners publish
// column k -> wave 0 picks the pivot among unused rows -> owners publish the pivot row -> everyone
// updates its NRxNR register tile.  Rows are pivoted logically (order[]).  Ends by writing the
// factor back to LDS and back-substituting inside wave 0.
template <int NR>
__device__ void lu_solve_regs(float* A, int ld, int n, float* x, int* order, float* colbuf /*[2][16*NR]*/,
                              float* rowbuf /*[16*NR]*/, int* pv /*[16*NR]*/, float* spiv /*[2]*/) {
  const int tid = threadIdx.x, ti = tid / kGrid, tj = tid % kGrid;
  float a[NR][NR];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      const int i = ti + kGrid * r, j = tj + kGrid * c;
      a[r][c] = (i < n && j <= n) ? A[i * ld + j] : 0.f;
    }
  for (int i = tid; i < kGrid * NR; i += kSolveThreads) pv[i] = (i < n) ? 0 : 1;
  __syncthreads();
#pragma unroll
  for (int c0 = 0; c0 < NR; ++c0) {
    for (int kk = 0; kk < kGrid; ++kk) {
      const int k = kGrid * c0 + kk;
      if (k >= n) break;  // uniform
      float* cb = colbuf + (k & 1) * kGrid * NR;
      if (tj == kk) {
#pragma unroll
        for (int r = 0; r < NR; ++r) cb[ti + kGrid * r] = a[r][c0];
      }
      __syncthreads();
      if (tid < 64) {
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < (kGrid * NR + 63) / 64; ++u) {
          const int i = tid + 64 * u;
          if (i < n) {
            const float v = pv[i] ? -1.f : fabsf(cb[i]);
            if (v > best) {
              best = v;
              bi = i;
            }
          }
        }
        const float mx = wave_max_fast(best);
        const int pi = wave_min_fast(best == mx ? bi : 0x7fffffff);   // smallest row index among the maxima
        if (tid == 0) {
          pv[pi] = 1;
          order[k] = pi;
          spiv[0] = __int_as_float(pi);
          spiv[1] = 1.f / cb[pi];
        }
      }
      __syncthreads();
      const int p = __float_as_int(spiv[0]);
      const float inv = spiv[1];
      if (ti == (p % kGrid)) {  // owners of the pivot row publish it
        const int rsel = p / kGrid;
#pragma unroll
        for (int c = c0; c < NR; ++c) {
          float v = a[0][c];
#pragma unroll
          for (int r = 1; r < NR; ++r) v = (r == rsel) ? a[r][c] : v;
          rowbuf[tj + kGrid * c] = v;
        }
      }
      __syncthreads();
      float rb[NR];
#pragma unroll
      for (int c = c0; c < NR; ++c) rb[c] = rowbuf[tj + kGrid * c];
      float lm[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {  // rows already used as a pivot (pv = 1, also beyond n) get multiplier 0
        const int i = ti + kGrid * r;
        lm[r] = pv[i] ? 0.f : -(cb[i] * inv);
      }
#pragma unroll
      for (int c = c0; c < NR; ++c) {
        const float rbc = (tj + kGrid * c > k) ? rb[c] : 0.f;  // columns <= k are finished
#pragma unroll
        for (int r = 0; r < NR; ++r) a[r][c] = fmaf(lm[r], rbc, a[r][c]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < NR; ++c) {
      const int i = ti + kGrid * r, j = tj + kGrid * c;
      if (i < n && j <= n) A[i * ld + j] = a[r][c];
    }
  __syncthreads();
  if (tid < 64) {  // back substitution: U's row k is physical row order[k]
    for (int k = n - 1; k >= 0; --k) {
      const float* row = A + order[k] * ld;
      float part = 0.f;
      for (int j = k + 1 + tid; j < n; j += 64) part = fmaf(row[j], x[j], part);
      const float tot = wave_sum_fast(part);
      if (tid == 0) x[k] = (row[n] - tot) / row[k];
    }
  }
  __syncthreads();
}

// --------------------------------------------------------------------------------------
// Blocked LDL^T for the bundle variants (32 <= n): the damped normal matrix is symmetric positive
// definite (PSD Gram matrix + positive damping), so the pivoted LU of tf.matrix_solve
// (bundlenet.py:267) reduces to elimination in natural order; same solution up to rounding
// (checked at 1e-4 by the parity tests).  Storage: lower triangle of A in rows 0..n-1, the right-hand
// side as ROW n -- eliminating it like any other row leaves w = D^-1 L^-1 b there, the right-hand
// side of the back substitution L^T x = w.  Right-looking, 16 columns per panel:
//   A  wave 0 factors the 16x16 diagonal block in registers (lane = row, pivot rows broadcast with
//      v_readlane) and publishes U11 and the reciprocal pivots;
//   B  one thread per remaining row eliminates its 16 panel entries against U11 (broadcast LDS
//      reads), keeps the multipliers in place and the unscaled entries d_j l_ij as W^T;
//   C  trailing update A22 -= L21 W (lower triangle + rhs row), one 4x4 register tile per thread.
// 3 barriers per panel instead of one per column, and no redundant upper-triangle work.
// --------------------------------------------------------------------------------------
constexpr int kPanel = 16;

__host__ __device__ inline int ldlt_ld(int n) {  // row stride: >= round16(n), odd multiple of 4 (conflict-free b128 by row)
  int ld = ((n + 15) & ~15) + 4;
  while ((ld & 7) != 4) ld += 4;
  return ld;
}
__host__ __device__ inline int ldlt_ldw(int n) { return ((n + 1 + 3) & ~3) + 4; }   // W^T row stride
__host__ __device__ inline int solve_scratch_floats(int P) {   // MLP partials (4096) or the LDL^T scratch, whichever is larger
  const int need = P >= 32 ? kPanel * ldlt_ldw(P) + 288 : 0;
  return need > 4096 ? ((need + 3) & ~3) : 4096;
}
__host__ __device__ inline size_t solve_big_floats(int P) { return (size_t)(P + 4) * ldlt_ld(P); }

__device__ __forceinline__ float rdlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __attribute__((always_inline)) void ldlt_solve_blocked(float* A, int ld, int n, float* x,
                                                                  float* scratch /* >= 16*ldw + 272 floats */) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int m = n + 1;                 // rows including the right-hand side
  const int ldw = ldlt_ldw(n);
  float* sWT = scratch;                // [16][ldw]
  float* sU = sWT + kPanel * ldw;      // [16][16]  upper part of the factored diagonal block
  float* sDinv = sU + kPanel * kPanel; // [16]
#ifdef BANET_TIMING
  __shared__ float sDbg[4];
  float dA = 0.f, dB = 0.f, dC = 0.f;
#endif
  for (int k0 = 0; k0 < n; k0 += kPanel) {
    const int nb = min(kPanel, n - k0);
    STICK(pa);
    // ---- A: diagonal block (wave 0) ------------------------------------------------------
    if (tid < 64) {
      const int row = lane & 15;
      float a[kPanel];
#pragma unroll
      for (int c = 0; c < kPanel; ++c) {
        const bool in = row < nb && c < nb;
        const int r = k0 + row, cc = k0 + c;
        const int idx = in ? (c <= row ? r * ld + cc : cc * ld + r) : 0;   // symmetric read from the lower triangle
        const float v = A[idx];
        a[c] = in ? v : (c == row ? 1.f : 0.f);                          // identity padding of a partial panel
      }
      float myinv = 1.f;
#pragma unroll
      for (int j = 0; j < kPanel; ++j) {
        // pivot row j: all broadcasts first, then the arithmetic (a v_readlane feeding the very next VALU instruction costs
        // wait states; batched, the 16 - j reads pipeline)
        float pr[kPanel];
#pragma unroll
        for (int c = j; c < kPanel; ++c) pr[c] = rdlane(a[c], j);
        __builtin_amdgcn_sched_barrier(0);
        const float inv = __builtin_amdgcn_rcpf(pr[j]);                  // v_rcp_f32: <= 1 ulp
        if (row == j) myinv = inv;
        const float l = row > j ? a[j] * inv : 0.f;
#pragma unroll
        for (int c = j + 1; c < kPanel; ++c) a[c] = fmaf(-l, pr[c], a[c]);
        a[j] = row > j ? l : a[j];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < kPanel) {
        sDinv[row] = myinv;
#pragma unroll
        for (int c = 0; c < kPanel; ++c) {
          if (c <= row) {
            if (row < nb && c < nb) A[(k0 + row) * ld + k0 + c] = a[c];
          } else {
            sU[row * kPanel + c] = a[c];
          }
        }
      }
    }
    __syncthreads();
    STICK(pb);
    // ---- B: panel rows below the block ---------------------------------------------------
    const int base = k0 + nb, mrem = m - base;
    for (int t = tid; t < mrem; t += kSolveThreads) {
      float* rowp = A + (base + t) * ld + k0;
      float a[kPanel];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(rowp + 4 * q);
        a[4 * q + 0] = (4 * q + 0 < nb) ? v.x : 0.f;
        a[4 * q + 1] = (4 * q + 1 < nb) ? v.y : 0.f;
        a[4 * q + 2] = (4 * q + 2 < nb) ? v.z : 0.f;
        a[4 * q + 3] = (4 * q + 3 < nb) ? v.w : 0.f;
      }
#pragma unroll
      for (int j = 0; j < kPanel; ++j) {
        sWT[j * ldw + t] = a[j];                        // d_j l_ij
        const float l = a[j] * sDinv[j];
        a[j] = l;
#pragma unroll
        for (int c = j + 1; c < kPanel; ++c) a[c] = fmaf(-l, sU[j * kPanel + c], a[c]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(rowp + 4 * q) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    }
    __syncthreads();
    STICK(pc);
    // ---- C: trailing update, 4x4 tiles of the lower triangle ---------------------------------
    const int mt = (mrem + 3) >> 2, ntiles = mt * (mt + 1) / 2;
    for (int t = tid; t < ntiles; t += kSolveThreads) {
      int I4 = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
      while ((I4 + 1) * (I4 + 2) / 2 <= t) ++I4;
      while (I4 * (I4 + 1) / 2 > t) --I4;
      const int C4 = t - I4 * (I4 + 1) / 2;
      float acc[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
      const float* Lp = A + (base + 4 * I4) * ld + k0;
      const float* Wp = sWT + 4 * C4;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        float4 Lr[4], Wj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) Lr[r] = *reinterpret_cast<const float4*>(Lp + r * ld + 4 * j4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) Wj[jj] = *reinterpret_cast<const float4*>(Wp + (4 * j4 + jj) * ldw);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lr[4] = {Lr[r].x, Lr[r].y, Lr[r].z, Lr[r].w};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            acc[r][0] = fmaf(lr[jj], Wj[jj].x, acc[r][0]);
            acc[r][1] = fmaf(lr[jj], Wj[jj].y, acc[r][1]);
            acc[r][2] = fmaf(lr[jj], Wj[jj].z, acc[r][2]);
            acc[r][3] = fmaf(lr[jj], Wj[jj].w, acc[r][3]);
          }
        }
      }
      float* Cp = A + (base + 4 * I4) * ld + base + 4 * C4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float4 cv = *reinterpret_cast<float4*>(Cp + r * ld);
        cv.x -= acc[r][0];
        cv.y -= acc[r][1];
        cv.z -= acc[r][2];
        cv.w -= acc[r][3];
        *reinterpret_cast<float4*>(Cp + r * ld) = cv;
      }
    }
    __syncthreads();
#ifdef BANET_TIMING
    {
      STICK(pd);
      dA += (float)(pb - pa);
      dB += (float)(pc - pb);
      dC += (float)(pd - pc);
    }
#endif
  }
  STICK(pe);
  // ---- back substitution L^T x = w (wave 0), panels from the last to the first ---------------
  if (tid < 64) {
    const int li = lane & 15, part = lane >> 4;
    const int npan = (n + kPanel - 1) / kPanel;
    for (int p = npan - 1; p >= 0; --p) {
      const int k0 = p * kPanel, nb = min(kPanel, n - k0), base = k0 + nb;
      // t_l = w_l - sum_{j >= base} L[j][k0 + l] x_j : 4 interleaved slices of j, one per 16-lane row
      float tsum = 0.f;
      {
        const float* Lc = A + k0 + (li < nb ? li : 0);
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int j = base + part;
        for (; j + 12 < n; j += 16) {   // 4 independent loads in flight per lane
          const float l0 = Lc[j * ld], l1 = Lc[(j + 4) * ld], l2 = Lc[(j + 8) * ld], l3 = Lc[(j + 12) * ld];
          t0 = fmaf(l0, x[j], t0);
          t1 = fmaf(l1, x[j + 4], t1);
          t2 = fmaf(l2, x[j + 8], t2);
          t3 = fmaf(l3, x[j + 12], t3);
        }
        for (; j < n; j += 4) t0 = fmaf(Lc[j * ld], x[j], t0);
        tsum = li < nb ? (t0 + t1) + (t2 + t3) : 0.f;
      }
      tsum += __shfl_xor(tsum, 16, 64);
      tsum += __shfl_xor(tsum, 32, 64);
      float tl = (li < nb ? A[n * ld + k0 + li] : 0.f) - tsum;
      float col[kPanel];   // column l of the block's unit lower triangle: L[k0 + i][k0 + l], i > l
#pragma unroll
      for (int i = 0; i < kPanel; ++i) col[i] = (i > li && i < nb) ? A[(k0 + i) * ld + k0 + li] : 0.f;
#pragma unroll
      for (int i = kPanel - 1; i >= 1; --i) {
        const float xi = rdlane(tl, i);      // final once rows > i have been applied
        tl = fmaf(-col[i], xi, tl);
      }
      if (lane < nb) x[k0 + lane] = tl;
    }
  }
  __syncthreads();
#ifdef BANET_TIMING
  {
    STICK(pf);
    if (tid == 0) {
      sDbg[0] = dA;
      sDbg[1] = dB;
      sDbg[2] = dC;
      sDbg[3] = (float)(pf - pe);
      x[n + 0] = dA; x[n + 1] = dB; x[n + 2] = dC; x[n + 3] = (float)(pf - pe);
    }
    __syncthreads();
  }
#endif
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

// --------------------------------------------------------------------------------------
// Conjugate gradients for the bundle variants' damped systems (round 3).  bundlenet.py:264-267 damps every diagonal entry but
// the last by lambda (diag + 1e-5) and solves with tf.matrix_solve; with the reference's l2_regularizer_base = 1000 the damped
// block A11 is, after Jacobi scaling, I + (correlation matrix) / lambda: condition number <= 1 + P / lambda, i.e. 1.07 at the
// lambda ~ 2000-40000 the layer runs at.  Jacobi-preconditioned CG then reaches float32 accuracy in 4-6 matrix-vector
// products (0.5 us each for the whole workgroup) where the blocked LDL^T is a 52 us chain of 9 panels x 3 barriers.  The
// undamped last coefficient is eliminated exactly by a Schur step: A11 [x1 | z] = [b1 | a12] (both right-hand sides in the same
// sweeps), x_last = (b_last - a12.x1) / (a_PP - a12.z), x = x1 - z x_last.  The matrix is only read; if the iteration
// does not reach the tolerance within kPcgMaxIt products (small lambda: trained weights, other data) or meets a non-positive
// curvature, the caller falls back to the LDL^T.  Fixed-order reductions: bit-reproducible.
// Thread layout: 4 threads per row (16-byte row reads, quad reduction by DPP); x and r live in the row owner's registers,
// only the search directions go through LDS.
// --------------------------------------------------------------------------------------
constexpr int kPcgMaxIt = 40;
constexpr float kPcgTol2 = 2.5e-14f;     // (1.6e-7)^2 on |r|^2 / |b|^2

// block-wide sums of NV values per thread (fixed order); sr: 16 * NV floats, a different buffer than the previous call's
template <int NV>
__device__ __forceinline__ void block_sum_n(float (&v)[NV], float* sr) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float t = wave_sum_fast(v[k]);
    if (lane == 0) sr[wv * NV + k] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < kSolveWaves; ++i) t += sr[i * NV + k];
    v[k] = t;
  }
}

__device__ bool pcg_schur_solve(const float* A, int ld, int n, bool undamped_last, float* x, float* scratch) {
  const int tid = threadIdx.x;
  const int n1 = undamped_last ? n - 1 : n;
  const int row = tid >> 2, q = tid & 3;
  const bool own = row < n1, owner = own && q == 0;
  const int n1p = (n1 + 3) & ~3;
  float* p0 = scratch;                 // search direction of right-hand side 0 (b1) ...
  float* p1 = p0 + n1p;                // ... and 1 (a12)
  float* srA = p1 + n1p;               // two reduction buffers used alternately (16 waves x 4 values)
  float* srB = srA + 4 * kSolveWaves;
  const float* Ar = A + (size_t)(own ? row : 0) * ld;
  float r0 = own ? A[(size_t)n * ld + row] : 0.f;                     // right-hand side: row n of the augmented matrix
  float r1 = (own && undamped_last) ? Ar[n - 1] : 0.f;               // a12
  const float a12 = r1;
  const float dinv = own ? 1.f / Ar[row] : 0.f;
  float x0 = 0.f, x1 = 0.f;
  float z0 = dinv * r0, z1 = dinv * r1;
  float red[4] = {owner ? r0 * z0 : 0.f, owner ? r1 * z1 : 0.f, owner ? r0 * r0 : 0.f, owner ? r1 * r1 : 0.f};
  block_sum_n<4>(red, srA);
  float rz0 = red[0], rz1 = red[1];
  const float bb0 = red[2], bb1 = red[3];
  if (owner) {
    p0[row] = z0;
    p1[row] = z1;
  }
  __syncthreads();
  bool done0 = !(bb0 > 0.f), done1 = !(bb1 > 0.f);                     // a zero right-hand side: x = 0
  bool bad = !(bb0 == bb0) || !(bb1 == bb1);                           // NaN input: let the LDL^T produce the documented result
  for (int it = 0; it < kPcgMaxIt && !(done0 && done1) && !bad; ++it) {
    // ---- A11 p for both right-hand sides: this thread's quarter of the row
    float a0 = 0.f, a1 = 0.f;
    if (own) {
      int c = 4 * q;
      for (; c + 3 < n1; c += 16) {
        const float4 av = *reinterpret_cast<const float4*>(Ar + c);
        const float4 u = *reinterpret_cast<const float4*>(p0 + c), v = *reinterpret_cast<const float4*>(p1 + c);
        a0 = fmaf(av.x, u.x, fmaf(av.y, u.y, fmaf(av.z, u.z, fmaf(av.w, u.w, a0))));
        a1 = fmaf(av.x, v.x, fmaf(av.y, v.y, fmaf(av.z, v.z, fmaf(av.w, v.w, a1))));
      }
      if (c < n1 && c + 3 >= n1)                                       // the row's ragged tail belongs to exactly one quarter
        for (int j = c; j < n1; ++j) {
          a0 = fmaf(Ar[j], p0[j], a0);
          a1 = fmaf(Ar[j], p1[j], a1);
        }
    }
    a0 += dpp_mov<kDppXor1>(a0);
    a1 += dpp_mov<kDppXor1>(a1);
    a0 += dpp_mov<kDppXor2>(a0);
    a1 += dpp_mov<kDppXor2>(a1);
    const float pi0 = owner ? p0[row] : 0.f, pi1 = owner ? p1[row] : 0.f;
    float pap[2] = {pi0 * a0, pi1 * a1};
    block_sum_n<2>(pap, srB);
    if ((!done0 && !(pap[0] > 0.f)) || (!done1 && !(pap[1] > 0.f))) {
      bad = true;                                                        // not positive definite in float32: not CG's job
      break;
    }
    const float al0 = done0 ? 0.f : rz0 / pap[0], al1 = done1 ? 0.f : rz1 / pap[1];
    if (owner) {
      x0 = fmaf(al0, pi0, x0);
      x1 = fmaf(al1, pi1, x1);
      r0 = fmaf(-al0, a0, r0);
      r1 = fmaf(-al1, a1, r1);
      z0 = dinv * r0;
      z1 = dinv * r1;
    }
    float rr[4] = {owner ? r0 * z0 : 0.f, owner ? r1 * z1 : 0.f, owner ? r0 * r0 : 0.f, owner ? r1 * r1 : 0.f};
    block_sum_n<4>(rr, srA);
    const float be0 = done0 ? 0.f : rr[0] / rz0, be1 = done1 ? 0.f : rr[1] / rz1;
    rz0 = done0 ? rz0 : rr[0];
    rz1 = done1 ? rz1 : rr[1];
    done0 = done0 || rr[2] <= kPcgTol2 * bb0;
    done1 = done1 || rr[3] <= kPcgTol2 * bb1;
    if (owner) {
      p0[row] = fmaf(be0, pi0, z0);
      p1[row] = fmaf(be1, pi1, z1);
    }
    __syncthreads();
  }
  const bool ok = done0 && done1 && !bad;
  if (ok) {
    if (undamped_last) {
      float d[2] = {owner ? a12 * x0 : 0.f, owner ? a12 * x1 : 0.f};
      block_sum_n<2>(d, srB);
      const float s = A[(size_t)(n - 1) * ld + n - 1] - d[1];
      const float t = A[(size_t)n * ld + n - 1] - d[0];
      const float xl = t / s;                                            // 0 / 0 when the coefficient is unobservable: NaN, as documented
      if (owner) x[row] = x0 - x1 * xl;
      if (tid == 0) x[n - 1] = xl;
    } else if (owner) {
      x[row] = x0;
    }
  }
  __syncthreads();
  return ok;
}

// BIG: the normal matrix lives in the caller's workspace instead of LDS (a separate instantiation, so that the
// common LDS-resident kernel keeps its register allocation)
template <bool BIG>
__global__ __launch_bounds__(kSolveThreads) void ba_solve_update_kernel(const SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int P = a.P, C = a.C, K = a.K;
  const bool blocked = P >= kGrid;                   // bundle variants: blocked LDL^T, rhs stored as row P
  const int ld = blocked ? ldlt_ld(P) : P + 2;
  const int arows = blocked ? P + 4 : P;
  // P too large for an LDS-resident matrix (K = 256 windows: P up to 6*7 + 256): the matrix lives in the caller's
  // workspace (L2-resident, 370 KB per window) and only the vectors / scratch stay in LDS.  Same code, global pointer.
  constexpr bool big = BIG;
  float* gA = big ? a.bigA + (size_t)b * solve_big_floats(P) : nullptr;
  float* sA = smem;                 // [arows][ld] augmented (unused when big)
  float* sX = sA + (big ? 0 : ((arows * ld + 3) & ~3));  // [P]  (every carve offset a multiple of 4 floats: float4 LDS accesses)
  float* sH0 = sX + ((P + 3) & ~3); // MLP ping
  float* sH1 = sH0 + 4 * C;         // MLP pong
  float* sAvg = sH1 + 4 * C;        // [C]
  float* sRed = sAvg + ((C + 3) & ~3);  // [8]
  float* sScal = sRed + 20;         // pivot reciprocal, -, flags
  float* sPart = sRed + 24;         // [solve_scratch_floats(P)] MLP partial sums / solver scratch
  int* sPerm = reinterpret_cast<int*>(sPart + solve_scratch_floats(P));  // [P]

  LmCtl* ctl = a.ctl ? a.ctl + b : nullptr;
  if (ctl && ctl->active == 0) return;

  STICK(tk0);
  const bool legacy = a.variant == BANET_LEGACY_LM || a.variant == BANET_LEGACY_FIXED;
  const int pairs = a.pairs;
  const float Nf = (float)a.N * (float)pairs;   // residual rows per window: N points x pairs target frames
  const float nval = a.nvalid[b];
  // ---- average residual -------------------------------------------------------------
  const float numvalid = Nf / nval;  // legacy/ba.py:257
  float ss = 0.f, sm = 0.f;
  for (int c = tid; c < C; c += kSolveThreads) {
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
  if (a.use_mlp && a.mlp_y != nullptr) {   // evaluated by the MLP role workgroup of this iteration's SYRK launch
    const float e = legacy ? 1.f : 2.f;
    lam = powf(nrm, e + a.mlp_y[b]);
  } else if (a.use_mlp) {
    __syncthreads();
    mlp_layer(sAvg, sH0, a.mlp.w[0], a.mlp.b[0], C, 2 * C, 0, sPart, sRed);
    mlp_layer(sH0, sH1, a.mlp.w[1], a.mlp.b[1], 2 * C, 4 * C, 0, sPart, sRed);
    mlp_layer(sH1, sH0, a.mlp.w[2], a.mlp.b[2], 4 * C, 2 * C, 0, sPart, sRed);
    mlp_layer(sH0, sH1, a.mlp.w[3], a.mlp.b[3], 2 * C, C, 0, sPart, sRed);
    mlp_layer(sH1, sH0, a.mlp.w[4], a.mlp.b[4], C, 1, 1, sPart, sRed);
    const float y = sH0[0];
    const float e = legacy ? 1.f : 2.f;                          // ba.py:274 / bundlenet.py:173,249
    lam = powf(nrm, e + y);
  } else {
    lam = powf(nrm, 2.f);                                        // legacy/ba.py:190
  }
  if (a.variant == BANET_BUNDLE) lam *= a.l2_base;               // bundlenet.py:252-253
  STICK(tk1);
  // ---- accept / terminate (legacy early termination) ----------------------------------
  if (ctl) {
    if (tid == 0) {
      int go = 1;
      if (ctl->pending) {
        if (!(avg_scalar < a.lm.residual_ratio * ctl->avg_prev)) {  // reject: legacy/ba.py:343-345
          for (int i = 0; i < 9; ++i) a.st.R[b * 9 + i] = ctl->Rprev[i];
          for (int i = 0; i < 3; ++i) a.st.T[b * 3 + i] = ctl->Tprev[i];
          ctl->uw = 0.f;
          ctl->ut = 0.f;
        }
        ctl->pending = 0;
      }
      // loop condition, legacy/ba.py:132-133
      if (!(a.st.iters[b] < a.max_iters && a.lm.angle_change < ctl->uw && a.lm.translation_change < ctl->ut)) {
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
  if (big) {
    for (int e = tid; e < P * P; e += kSolveThreads) {
      const int i = e / P, j = e - i * P;
      float v = A_g[e];
      if (i == j && !(a.variant == BANET_BUNDLE && i == P - 1)) v = v + (v + 1e-5f) * lam;
      gA[i * ld + j] = v;
    }
    for (int i = tid; i < P; i += kSolveThreads) gA[P * ld + i] = a.Atb[(size_t)b * P + i];
    __threadfence_block();
  } else {
    for (int e = tid; e < P * P; e += kSolveThreads) {
      const int i = e / P, j = e - i * P;
      float v = A_g[e];
      if (i == j && !(a.variant == BANET_BUNDLE && i == P - 1)) v = v + (v + 1e-5f) * lam;
      sA[i * ld + j] = v;
    }
    for (int i = tid; i < P; i += kSolveThreads) sA[blocked ? P * ld + i : i * ld + P] = a.Atb[(size_t)b * P + i];
  }
  __syncthreads();
  STICK(tk2);
  // ---- solve --------------------------------------------------------------------------
  if (legacy && P == 6) {
    if (tid == 0) {
      float rhs[6];
      for (int i = 0; i < 6; ++i) rhs[i] = sA[i * ld + P];
      if (a.lm.solver == BANET_SOLVER_INVERSE)
        inverse_solve_small(sA, ld, rhs, sX);     // `qr = False`, legacy/ba.py:203,290
      else
        qr_solve_small(sA, ld, rhs, 6, sX);
    }
    __syncthreads();
  } else {
    float* sCol = sPart;                // the MLP scratch is free by now (4096 floats)
    if (P <= kGrid - 1) {               // pose-only variants: pivoted LU as tf.matrix_solve
      lu_solve_regs<1>(sA, ld, P, sX, sPerm, sCol, sCol + 2 * kGrid, reinterpret_cast<int*>(sCol + 3 * kGrid), sScal);
    } else if constexpr (big) {
      ldlt_solve_blocked(gA, ld, P, sX, sCol);    // matrix in global memory (workgroup-private, L2)
    } else {
      // matrix in LDS: conjugate gradients on the damped block first (reads the matrix only), the LDL^T when they do not
      // converge (flags bit 23: LDL^T only, A/B and parity tests)
      bool cg_ok = false;
      // pcg_schur_solve covers 4 threads per row and needs 2 n1p + 8 kSolveWaves floats of scratch: larger systems (none is
      // LDS-resident today: solve_lds_bytes stops at P ~ 190) go straight to the factorisation instead of dropping rows
      const bool cg_fits = 4 * P <= kSolveThreads && 2 * ((P + 3) & ~3) + 8 * kSolveWaves <= solve_scratch_floats(P);
      if (cg_fits && !(a.flags & (1 << 23))) cg_ok = pcg_schur_solve(smem, ld, P, a.variant == BANET_BUNDLE, sX, sCol);
      if (!cg_ok) ldlt_solve_blocked(smem, ld, P, sX, sCol);
    }
  }
  STICK(tk3);
  // ---- update -------------------------------------------------------------------------
  for (int k = tid; k < P; k += kSolveThreads) a.st.delta[(size_t)b * P + k] = sX[k];
  if (a.queue != nullptr && tid < a.nqueue) a.queue[b * a.nqueue + tid] = 0;   // tile queue of the next gather
  for (int k = tid; k < K; k += kSolveThreads) a.st.Wc[(size_t)b * K + k] += sX[6 * pairs + k];   // bundlenet.py:276
  if (tid < pairs) {   // one thread per target frame's pose
    const int pb = b * pairs + tid;
    const float* sx = sX + 6 * tid;
    float w[3] = {sx[0], sx[1], sx[2]}, t[3] = {sx[3], sx[4], sx[5]};
    float Rw[9], V[9], Ro[9], To[3], Rn[9], Tn[3];
    for (int i = 0; i < 9; ++i) Ro[i] = a.st.R[pb * 9 + i];
    for (int i = 0; i < 3; ++i) To[i] = a.st.T[pb * 3 + i];
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
    for (int i = 0; i < 9; ++i) a.st.R[pb * 9 + i] = Rn[i];
    for (int i = 0; i < 3; ++i) a.st.T[pb * 3 + i] = Tn[i];
    if (tid == 0) {
      a.st.iters[b] += 1;
      a.st.ratio[b] = (a.variant == BANET_LEGACY_FIXED) ? nval / Nf : numvalid;   // ba.py:214 / :344
      a.st.lambda_out[b] = lam;
    }
  }
#ifdef BANET_TIMING
  __syncthreads();
  if (tid == 0) {
    const unsigned long long tk4 = stick();
    float* dp = a.st.delta + (size_t)b * P;
    dp[0] = (float)(tk1 - tk0);  // avg + MLP
    dp[1] = (float)(tk2 - tk1);  // accept logic + damping/load
    dp[2] = (float)(tk3 - tk2);  // LU + back substitution
    dp[3] = (float)(tk4 - tk3);  // update
    dp[4] = sX[P]; dp[5] = sX[P + 1]; dp[6] = sX[P + 2]; dp[7] = sX[P + 3];
  }
#endif
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

size_t solve_lds_bytes(int P, int C, bool big) {
  const bool blocked = P >= kGrid;
  const int ld = blocked ? ldlt_ld(P) : P + 2, arows = blocked ? P + 4 : P;
  const size_t fl = (size_t)(big ? 0 : ((arows * ld + 3) & ~3)) + ((P + 3) & ~3) + 8 * C + ((C + 3) & ~3) + 24 +
                    solve_scratch_floats(P) + P + 8;
  return fl * sizeof(float);
}

// bytes of caller workspace the solve needs for its matrix (0: it fits in LDS)
size_t solve_big_bytes(int B, int P, int C) {
  if (solve_lds_bytes(P, C, false) <= 160 * 1024) return 0;
  return (size_t)B * solve_big_floats(P) * sizeof(float);
}

int launch_solve(const SolveArgs& a, hipStream_t s) {
  const bool need_big = solve_lds_bytes(a.P, a.C, false) > 160 * 1024;
  if (need_big && a.bigA == nullptr) return BANET_ERR_UNSUPPORTED;   // only banet_lm_level_f32 has a workspace for it
  const size_t lds = solve_lds_bytes(a.P, a.C, need_big);
  if (lds > 160 * 1024) return BANET_ERR_UNSUPPORTED;
  if (need_big) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)ba_solve_update_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ba_solve_update_kernel<true>, dim3(a.B), dim3(kSolveThreads), lds, s, a);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)ba_solve_update_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ba_solve_update_kernel<false>, dim3(a.B), dim3(kSolveThreads), lds, s, a);
  }
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

// --------------------------------------------------------------------------------------
// banet_spd_solve_f32: x = A^-1 b for B symmetric positive definite systems (32 <= P <= ~190), the blocked LDL^T of the
// update kernel as an op of its own.  Used by the backward of the dense layer (banet_amd/dense_train.py): the forward
// solution of the damped system is recomputed and the implicit-function gradient lam = A^-T g is a second call with the
// same matrix -- torch.linalg.solve costs ~2 ms per iteration there (rocSOLVER getrf + per-item trsv launches).
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSolveThreads) void spd_solve_kernel(const float* __restrict__ A, const float* __restrict__ rhs,
                                                                  float* __restrict__ x, int P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ld = ldlt_ld(P);
  float* sA = smem;                                   // [(P + 4)][ld], right-hand side as row P
  float* sX = sA + (((P + 4) * ld + 3) & ~3);         // [P + 8]
  float* sScr = sX + ((P + 8 + 3) & ~3);              // [solve_scratch_floats(P)]
  const float* A_g = A + (size_t)b * P * P;
  for (int e = tid; e < P * P; e += kSolveThreads) {
    const int i = e / P, j = e - i * P;
    sA[i * ld + j] = A_g[e];
  }
  for (int i = tid; i < P; i += kSolveThreads) sA[P * ld + i] = rhs[(size_t)b * P + i];
  __syncthreads();
  ldlt_solve_blocked(smem, ld, P, sX, sScr);
  for (int k = tid; k < P; k += kSolveThreads) x[(size_t)b * P + k] = sX[k];
}

static size_t spd_solve_lds_bytes(int P) {
  const size_t fl = (size_t)(((P + 4) * ldlt_ld(P) + 3) & ~3) + ((P + 8 + 3) & ~3) + (size_t)solve_scratch_floats(P);
  return fl * sizeof(float);
}

bool spd_solve_fits(int P) { return P >= kGrid && spd_solve_lds_bytes(P) <= 160 * 1024; }

int launch_spd_solve(const float* A, const float* rhs, float* x, int B, int P, hipStream_t s) {
  if (P < kGrid) return BANET_ERR_UNSUPPORTED;                 // the blocked factorisation wants at least two panels
  const size_t lds = spd_solve_lds_bytes(P);
  if (lds > 160 * 1024) return BANET_ERR_UNSUPPORTED;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)spd_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(spd_solve_kernel, dim3(B), dim3(kSolveThreads), lds, s, A, rhs, x, P);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

void launch_ctl_init(LmCtl* ctl, int32_t* iters, int B, hipStream_t s) {
  hipLaunchKernelGGL(lm_ctl_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ctl, iters, B);
}

void launch_zero_iters(int32_t* iters, int B, hipStream_t s) {
  hipLaunchKernelGGL(zero_iters_kernel, dim3((B + 63) / 64), dim3(64), 0, s, iters, B);
}

}  // namespace banet
