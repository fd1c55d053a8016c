// Backward of the SMALL part of one BundleIteration / CameraIteration (bundlenet.py:165-190, 241-276 after the EquationConstruction
// op) as HIP kernels -- what tf.gradients derives for: avg = sum|d| / N -> lambda MLP (five k = 1 convolutions, selu x 4, tanh)
// -> lambda = l2 ||avg||^(2 + y) -> damping (last coefficient undamped in the bundle variant, bundlenet.py:264-266) ->
// tf.matrix_solve -> SE(3) / W update (bundlenet.py:268-276, AngleaAxisRotation :17-37, VMatrix :39-46).  Until round 5 this was a
// torch graph of ~150 launches per iteration (banet_amd/dense_train.py::_small_grads), eager above 8 windows.
//
// Given the state before the update (R, T), the forward's saved solution `delta`, the assembly outputs (AtA, Atb, sum|d|) and the
// upstream gradients of the updated state (gR', gT', gW'):
//   small_pre_kernel    one workgroup per window: lambda MLP forward (activations kept), lambda; the adjoint of the SE(3) / W update
//                       -> dL/d(R, T) (direct part) and dL/dsol; the damped matrix A and the right-hand side dL/dsol for the solve
//   spd_solve_kernel    (solve.hip) lam_adj = A^-1 dL/dsol          [P < 32: a Cholesky in small_post_kernel instead]
//   small_post_kernel   one workgroup per window: dL/dAtb = lam_adj, dL/dA = -lam_adj sol^T (implicit function theorem, A = A^T),
//                       dL/dAtA (diagonal x (1 + lambda) where damped), dL/dlambda -> dL/dy, dL/d||avg|| -> MLP backward (deltas
//                       kept) -> dL/d avg -> dL/d sum|d|
//   small_wgrad_kernel  the ten lambda-weight gradients, summed over the windows in window order (fixed order: bit-reproducible),
//                       ACCUMULATED into the caller's buffers (the iterations of a level share them)
// dL/dWc = gW' (W' = W + sol): the caller keeps it.
#include <algorithm>

#include "kernels.hpp"
#include "mlp.hpp"

namespace banet {

namespace {

constexpr int kSsThreads = 512;
constexpr int kSsWaves = kSsThreads / 64;

// per-window activation / delta record: h1 [2C] | h2 [4C] | h3 [2C] | h4 [C] | y [1] (+ pad) ; the input avg [C] is kept in front
__host__ __device__ inline int ss_act_floats(int C) { return ((10 * C + 1 + 3) & ~3); }   // avg C | h1 2C | h2 4C | h3 2C | h4 C | y
__host__ __device__ inline int ss_off(int C, int l) {   // offset of layer l's INPUT inside the record (l = 0..4), l = 5: y
  const int o[6] = {0, C, 3 * C, 7 * C, 9 * C, 10 * C};
  return o[l];
}
__host__ __device__ inline int ss_nin(int C, int l) {
  const int n[5] = {C, 2 * C, 4 * C, 2 * C, C};
  return n[l];
}
__host__ __device__ inline int ss_nout(int C, int l) {
  const int n[5] = {2 * C, 4 * C, 2 * C, C, 1};
  return n[l];
}

struct SmallArgs {
  int B, N, C, K, P, pairs, camera;
  float l2_base;
  banet_mlp_t mlp, gmlp;
  const float* AtA;
  const float* Atb;
  const float* absres;
  const float* delta;
  const float* R;
  const float* T;
  const float* gR;
  const float* gT;
  const float* gW;
  float* gAtA;
  float* gAtb;
  float* gabs;
  float* dR;
  float* dT;
  float* Ad;      // [B][P][P]  damped matrix
  float* rhs;     // [B][P]     dL/dsol
  float* ladj;    // [B][P]     A^-1 dL/dsol
  float* scal;    // [B][4]     lambda, ||avg||, y
  float* act;     // [B][ss_act_floats]
  float* dlt;     // [B][ss_act_floats]  deltas dL/dz_l at the offsets of the layers' OUTPUTS (ss_off(l + 1))
};

// adjoint of (R', T') = (exp(w) R, V(w) t + exp(w) T) for one target frame, in double (a handful of operations per window; the
// small-angle differences cancel badly in float32): -> gRo [9], gTo [3], gsol [6] = dL/d(w, t)
__device__ void se3_update_adjoint(const float* sol6, const float* Ro_, const float* To_, const float* gRn_, const float* gTn_, float* gRo,
                                   float* gTo, float* gsol) {
  double w[3] = {sol6[0], sol6[1], sol6[2]}, t[3] = {sol6[3], sol6[4], sol6[5]};
  double Ro[9], To[3], gRn[9], gTn[3];
  for (int i = 0; i < 9; ++i) Ro[i] = Ro_[i], gRn[i] = gRn_[i];
  for (int i = 0; i < 3; ++i) To[i] = To_[i], gTn[i] = gTn_[i];
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double thc = fmax(th, 1e-6);                            // bundlenet.py:19 (clip_by_value)
  const double kx = w[0] / thc, ky = w[1] / thc, kz = w[2] / thc;
  const double c = cos(thc), s = sin(thc), oc = 1.0 - c;
  double Rw[9];
  Rw[0] = c + kx * kx * oc;
  Rw[1] = kx * ky * oc - kz * s;
  Rw[2] = ky * s + kx * kz * oc;
  Rw[3] = kz * s + kx * ky * oc;
  Rw[4] = c + ky * ky * oc;
  Rw[5] = -kx * s + ky * kz * oc;
  Rw[6] = -ky * s + kx * kz * oc;
  Rw[7] = kx * s + ky * kz * oc;
  Rw[8] = c + kz * kz * oc;
  double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Kx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, K2[9], av = 0, bq = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K2[i * 3 + j] = Kx[i * 3] * Kx[j] + Kx[i * 3 + 1] * Kx[3 + j] + Kx[i * 3 + 2] * Kx[6 + j];
  const bool vok = th > 0.0;
  if (vok) {
    av = (1.0 - cos(th)) / (th * th);
    bq = (th - sin(th)) / (th * th * th);
    for (int i = 0; i < 9; ++i) V[i] += av * Kx[i] + bq * K2[i];
  }
  // Rn = Rw Ro, Tn = V t + Rw To
  double gRw[9], gV[9], gt[3];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) {
      gRw[r * 3 + cc] = gRn[r * 3] * Ro[cc * 3] + gRn[r * 3 + 1] * Ro[cc * 3 + 1] + gRn[r * 3 + 2] * Ro[cc * 3 + 2] + gTn[r] * To[cc];
      gRo[r * 3 + cc] = (float)(Rw[r] * gRn[cc] + Rw[3 + r] * gRn[3 + cc] + Rw[6 + r] * gRn[6 + cc]);
      gV[r * 3 + cc] = gTn[r] * t[cc];
    }
  for (int r = 0; r < 3; ++r) {
    gTo[r] = (float)(Rw[r] * gTn[0] + Rw[3 + r] * gTn[1] + Rw[6 + r] * gTn[2]);
    gt[r] = V[r] * gTn[0] + V[3 + r] * gTn[1] + V[6 + r] * gTn[2];
  }
  // ---- exp(w), in reverse
  double gw[3] = {0, 0, 0};
  {
    double gc = gRw[0] + gRw[4] + gRw[8];
    const double goc = gRw[0] * kx * kx + (gRw[1] + gRw[3]) * kx * ky + (gRw[2] + gRw[6]) * kx * kz + gRw[4] * ky * ky +
                       (gRw[5] + gRw[7]) * ky * kz + gRw[8] * kz * kz;
    const double gs = -kz * gRw[1] + ky * gRw[2] + kz * gRw[3] - kx * gRw[5] - ky * gRw[6] + kx * gRw[7];
    const double gkx = oc * (2.0 * kx * gRw[0] + ky * (gRw[1] + gRw[3]) + kz * (gRw[2] + gRw[6])) + s * (gRw[7] - gRw[5]);
    const double gky = oc * (kx * (gRw[1] + gRw[3]) + 2.0 * ky * gRw[4] + kz * (gRw[5] + gRw[7])) + s * (gRw[2] - gRw[6]);
    const double gkz = oc * (kx * (gRw[2] + gRw[6]) + ky * (gRw[5] + gRw[7]) + 2.0 * kz * gRw[8]) + s * (gRw[3] - gRw[1]);
    gc -= goc;
    double gthc = -s * gc + c * gs;
    gw[0] += gkx / thc;
    gw[1] += gky / thc;
    gw[2] += gkz / thc;
    gthc -= (gkx * w[0] + gky * w[1] + gkz * w[2]) / (thc * thc);
    if (th >= 1e-6) {                                           // (the clamp passes the gradient where it is inactive)
      gw[0] += gthc * w[0] / th;
      gw[1] += gthc * w[1] / th;
      gw[2] += gthc * w[2] / th;
    }
  }
  // ---- V(w), in reverse
  if (vok) {
    double ga = 0, gb = 0, gK[9];
    for (int i = 0; i < 9; ++i) {
      ga += gV[i] * Kx[i];
      gb += gV[i] * K2[i];
    }
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        double v = av * gV[r * 3 + cc];
        for (int k = 0; k < 3; ++k) v += bq * (gV[r * 3 + k] * Kx[cc * 3 + k] + Kx[k * 3 + r] * gV[k * 3 + cc]);   // gK2 K^T + K^T gK2
        gK[r * 3 + cc] = v;
      }
    gw[0] += gK[7] - gK[5];
    gw[1] += gK[2] - gK[6];
    gw[2] += gK[3] - gK[1];
    const double sn = sin(th), cs = cos(th);
    const double da = (th * sn - 2.0 * (1.0 - cs)) / (th * th * th);
    const double db = (th * (1.0 - cs) - 3.0 * (th - sn)) / (th * th * th * th);
    const double gth = ga * da + gb * db;
    gw[0] += gth * w[0] / th;
    gw[1] += gth * w[1] / th;
    gw[2] += gth * w[2] / th;
  }
  for (int i = 0; i < 3; ++i) {
    gsol[i] = (float)gw[i];
    gsol[3 + i] = (float)gt[i];
  }
}

__global__ __launch_bounds__(kSsThreads) void small_pre_kernel(const SmallArgs a) {
  __shared__ __attribute__((aligned(16))) float sAvg[256];
  __shared__ __attribute__((aligned(16))) float sH[2][1024];
  __shared__ __attribute__((aligned(16))) float sPart[4096];
  __shared__ float sRed[32];
  const int b = blockIdx.x, tid = threadIdx.x, C = a.C, P = a.P, K = a.K, pairs = a.pairs;
  const float Nf = (float)a.N * (float)pairs;
  float* __restrict__ act = a.act + (size_t)b * ss_act_floats(C);
  float ss = 0.f;
  for (int c = tid; c < C; c += kSsThreads) {
    const float v = a.absres[(size_t)b * C + c] / Nf;             // bundlenet.py:243 (reduce_mean over the residual rows)
    sAvg[c] = v;
    act[c] = v;
    ss += v * v;
  }
  const float nrm = sqrtf(block_sum_t<kSsThreads>(ss, sRed));
  __syncthreads();
  // lambda MLP forward, every layer's output kept (bundlenet.py:168-172 / :244-248)
  const float* in = sAvg;
  for (int l = 0; l < 5; ++l) {
    float* out = sH[l & 1];
    mlp_layer_t<kSsThreads>(in, out, a.mlp.w[l], a.mlp.b[l], ss_nin(C, l), ss_nout(C, l), l == 4 ? 1 : 0, sPart, sRed);
    for (int o = tid; o < ss_nout(C, l); o += kSsThreads) act[ss_off(C, l + 1) + o] = out[o];
    in = out;
  }
  const float y = sH[0][0];                                        // layer 4 wrote sH[4 & 1]
  float lam = powf(nrm, 2.f + y);                                  // bundlenet.py:173 / :249
  if (!a.camera) lam *= a.l2_base;                                 // :252-253
  if (tid == 0) {
    float* sc = a.scal + (size_t)b * 4;
    sc[0] = lam;
    sc[1] = nrm;
    sc[2] = y;
    sc[3] = 0.f;
  }
  // the damped matrix (bundlenet.py:181-182 / :264-266) and the right-hand side dL/dsol
  const float* __restrict__ A_g = a.AtA + (size_t)b * P * P;
  float* __restrict__ Ad = a.Ad + (size_t)b * P * P;
  for (int e = tid; e < P * P; e += kSsThreads) {
    const int i = e / P, j = e - i * P;
    float v = A_g[e];
    if (i == j && (a.camera || i != P - 1)) v = v + (v + 1e-5f) * lam;
    Ad[e] = v;
  }
  float* __restrict__ rhs = a.rhs + (size_t)b * P;
  for (int k = tid; k < K; k += kSsThreads) rhs[6 * pairs + k] = a.gW[(size_t)b * K + k];      // W' = W + sol
  if (tid < pairs) {
    const int pb = b * pairs + tid;
    float gRo[9], gTo[3], gsol[6];
    se3_update_adjoint(a.delta + (size_t)b * P + 6 * tid, a.R + (size_t)pb * 9, a.T + (size_t)pb * 3, a.gR + (size_t)pb * 9,
                       a.gT + (size_t)pb * 3, gRo, gTo, gsol);
    for (int i = 0; i < 9; ++i) a.dR[(size_t)pb * 9 + i] = gRo[i];
    for (int i = 0; i < 3; ++i) a.dT[(size_t)pb * 3 + i] = gTo[i];
    for (int i = 0; i < 6; ++i) rhs[6 * tid + i] = gsol[i];
  }
}

// g_in[i] = sum_o W[i][o] d[o] (W [nin][nout] row-major), one wave per row, lanes over the outputs: coalesced weight reads
__device__ void mlp_layer_backward(const float* __restrict__ Wt, int nin, int nout, const float* dl /*LDS [nout]*/, float* gin /*LDS [nin]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = w; i < nin; i += kSsWaves) {
    float s = 0.f;
    for (int o = lane; o < nout; o += 64) s = fmaf(Wt[(size_t)i * nout + o], dl[o], s);
    s = wave_sum(s);
    if (lane == 0) gin[i] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kSsThreads) void small_post_kernel(const SmallArgs a) {
  __shared__ float sG[2][1024];      // gradient w.r.t. a layer's output -> delta, ping-pong
  __shared__ float sRed[32];
  __shared__ float sL[1024];         // P < 32: the matrix for the in-kernel Cholesky (P <= 31)
  __shared__ float sX[64];
  const int b = blockIdx.x, tid = threadIdx.x, C = a.C, P = a.P, pairs = a.pairs;
  const float Nf = (float)a.N * (float)pairs;
  const float* __restrict__ sc = a.scal + (size_t)b * 4;
  const float lam = sc[0], nrm = sc[1], y = sc[2];
  const float* __restrict__ sol = a.delta + (size_t)b * P;
  float* __restrict__ ladj = a.ladj + (size_t)b * P;
  if (P < 32) {     // pose only (P = 6): lam_adj = A^-1 dL/dsol by a Cholesky factorisation, one thread
    const float* __restrict__ Ad = a.Ad + (size_t)b * P * P;
    for (int e = tid; e < P * P; e += kSsThreads) sL[e] = Ad[e];
    for (int i = tid; i < P; i += kSsThreads) sX[i] = a.rhs[(size_t)b * P + i];
    __syncthreads();
    if (tid == 0) {
      for (int j = 0; j < P; ++j) {
        float d = sL[j * P + j];
        for (int k = 0; k < j; ++k) d -= sL[j * P + k] * sL[j * P + k];
        d = sqrtf(d);
        sL[j * P + j] = d;
        for (int i = j + 1; i < P; ++i) {
          float v = sL[i * P + j];
          for (int k = 0; k < j; ++k) v -= sL[i * P + k] * sL[j * P + k];
          sL[i * P + j] = v / d;
        }
      }
      for (int i = 0; i < P; ++i) {
        float v = sX[i];
        for (int k = 0; k < i; ++k) v -= sL[i * P + k] * sX[k];
        sX[i] = v / sL[i * P + i];
      }
      for (int i = P - 1; i >= 0; --i) {
        float v = sX[i];
        for (int k = i + 1; k < P; ++k) v -= sL[k * P + i] * sX[k];
        sX[i] = v / sL[i * P + i];
      }
      for (int i = 0; i < P; ++i) ladj[i] = sX[i];
    }
    __syncthreads();
  }
  // dL/dAtb = lam_adj; dL/dA = -lam_adj sol^T; A_ii = AtA_ii + (AtA_ii + 1e-5) lambda where damped
  const float* __restrict__ A_g = a.AtA + (size_t)b * P * P;
  float* __restrict__ gA = a.gAtA + (size_t)b * P * P;
  float gl = 0.f;
  for (int e = tid; e < P * P; e += kSsThreads) {
    const int i = e / P, j = e - i * P;
    float v = -ladj[i] * sol[j];
    if (i == j && (a.camera || i != P - 1)) {
      gl = fmaf(v, A_g[e] + 1e-5f, gl);
      v = v * (1.f + lam);
    }
    gA[e] = v;
  }
  for (int i = tid; i < P; i += kSsThreads) a.gAtb[(size_t)b * P + i] = ladj[i];
  const float glam = block_sum_t<kSsThreads>(gl, sRed);
  // lambda = l2 ||avg||^(2 + y)
  const float gy = glam * lam * logf(nrm);
  const float gnrm = glam * lam * (2.f + y) / nrm;
  const float* __restrict__ act = a.act + (size_t)b * ss_act_floats(C);
  float* __restrict__ dlt = a.dlt + (size_t)b * ss_act_floats(C);
  // MLP backward: delta_l = dL/dz_l; selu'(z) from the output h: h > 0 ? scale : h + scale alpha; tanh' = 1 - y^2
  __syncthreads();
  if (tid == 0) {
    const float d5 = gy * (1.f - y * y);
    sG[0][0] = d5;
    dlt[ss_off(C, 5)] = d5;
  }
  __syncthreads();
  for (int l = 4; l >= 0; --l) {
    const float* dl = sG[(4 - l) & 1];
    float* gin = sG[(5 - l) & 1];
    mlp_layer_backward(a.mlp.w[l], ss_nin(C, l), ss_nout(C, l), dl, gin);
    if (l > 0) {
      for (int i = tid; i < ss_nin(C, l); i += kSsThreads) {
        const float h = act[ss_off(C, l) + i];
        const float d = gin[i] * (h > 0.f ? kSeluScale : h + kSeluScale * kSeluAlpha);
        gin[i] = d;
        dlt[ss_off(C, l) + i] = d;
      }
    } else {
      for (int c = tid; c < C; c += kSsThreads) {
        const float g = gin[c] + gnrm * act[c] / nrm;        // d||avg|| / d avg = avg / ||avg||
        a.gabs[(size_t)b * C + c] = g / Nf;
      }
    }
    __syncthreads();
  }
}

// gW_l[i][o] += sum_b h_{l-1}[b][i] delta_l[b][o]; gb_l[o] += sum_b delta_l[b][o]; one thread per weight, windows in order
__global__ void small_wgrad_kernel(const SmallArgs a) {
  const int C = a.C, l = blockIdx.y;
  const int nin = ss_nin(C, l), nout = ss_nout(C, l);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (nin + 1) * nout) return;
  const int i = e / nout, o = e - i * nout;          // i == nin: the bias
  const int rs = ss_act_floats(C);
  const float* __restrict__ hp = a.act + ss_off(C, l) + (i < nin ? i : 0);
  const float* __restrict__ dp = a.dlt + ss_off(C, l + 1) + o;
  float s = 0.f;
  for (int b = 0; b < a.B; ++b) s = fmaf(i < nin ? hp[(size_t)b * rs] : 1.f, dp[(size_t)b * rs], s);
  if (i < nin)
    const_cast<float*>(a.gmlp.w[l])[(size_t)i * nout + o] += s;      // (banet_mlp_t's members are const float*: the caller's gradient buffers)
  else
    const_cast<float*>(a.gmlp.b[l])[o] += s;
}

struct SsPlan {
  size_t off_Ad, off_rhs, off_ladj, off_scal, off_act, off_dlt, bytes;
};
void ss_plan(int B, int C, int P, SsPlan* pl) {
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  pl->off_Ad = take((size_t)B * P * P * 4);
  pl->off_rhs = take((size_t)B * P * 4);
  pl->off_ladj = take((size_t)B * P * 4);
  pl->off_scal = take((size_t)B * 4 * 4);
  pl->off_act = take((size_t)B * ss_act_floats(C) * 4);
  pl->off_dlt = take((size_t)B * ss_act_floats(C) * 4);
  pl->bytes = o;
}

}  // namespace

bool small_step_supported(int variant, int B, int N, int C, int K, int pairs) {
  if (B <= 0 || N <= 0 || pairs < 1) return false;
  if (C < 1 || C > 256) return false;
  const int P = 6 * pairs + K;
  if (variant == BANET_BUNDLE) {
    if (K < 1) return false;
  } else if (variant == BANET_BUNDLE_CAMERA) {
    if (K != 0) return false;
  } else {
    return false;
  }
  if (P >= 32) return spd_solve_fits(P);      // the solve runs on spd_solve_kernel: its matrix must fit the LDS
  return true;
}

size_t small_step_workspace_bytes(int variant, int B, int N, int C, int K, int pairs) {
  if (!small_step_supported(variant, B, N, C, K, pairs)) return 0;
  SsPlan pl;
  ss_plan(B, C, 6 * pairs + K, &pl);
  return pl.bytes;
}

int launch_small_step_adjoint(int variant, int B, int N, int C, int K, int pairs, float l2_base, const banet_mlp_t* mlp, const float* AtA,
                              const float* Atb, const float* absres, const float* delta, const float* R, const float* T, const float* gR,
                              const float* gT, const float* gW, float* gAtA, float* gAtb, float* gabs, float* dR, float* dT,
                              const banet_mlp_t* gmlp, void* ws, hipStream_t s) {
  if (!small_step_supported(variant, B, N, C, K, pairs)) return BANET_ERR_UNSUPPORTED;
  const int P = 6 * pairs + K;
  SsPlan pl;
  ss_plan(B, C, P, &pl);
  char* base = static_cast<char*>(ws);
  SmallArgs a;
  a.B = B, a.N = N, a.C = C, a.K = K, a.P = P, a.pairs = pairs, a.camera = variant == BANET_BUNDLE_CAMERA ? 1 : 0;
  a.l2_base = l2_base;
  a.mlp = *mlp;
  a.gmlp = *gmlp;
  a.AtA = AtA, a.Atb = Atb, a.absres = absres, a.delta = delta, a.R = R, a.T = T, a.gR = gR, a.gT = gT, a.gW = gW;
  a.gAtA = gAtA, a.gAtb = gAtb, a.gabs = gabs, a.dR = dR, a.dT = dT;
  a.Ad = reinterpret_cast<float*>(base + pl.off_Ad);
  a.rhs = reinterpret_cast<float*>(base + pl.off_rhs);
  a.ladj = reinterpret_cast<float*>(base + pl.off_ladj);
  a.scal = reinterpret_cast<float*>(base + pl.off_scal);
  a.act = reinterpret_cast<float*>(base + pl.off_act);
  a.dlt = reinterpret_cast<float*>(base + pl.off_dlt);
  hipLaunchKernelGGL(small_pre_kernel, dim3(B), dim3(kSsThreads), 0, s, a);
  if (P >= 32) {
    const int rc = launch_spd_solve(a.Ad, a.rhs, a.ladj, B, P, s);
    if (rc != BANET_OK) return rc;
  }
  hipLaunchKernelGGL(small_post_kernel, dim3(B), dim3(kSsThreads), 0, s, a);
  const int maxw = (4 * C + 1) * 2 * C;        // the largest layer: (4C + 1) x 2C and (2C + 1) x 4C
  const int maxw2 = (2 * C + 1) * 4 * C;
  hipLaunchKernelGGL(small_wgrad_kernel, dim3((std::max(maxw, maxw2) + 255) / 256, 5), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
