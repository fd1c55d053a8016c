// Shared device helpers for the gfx950 BA kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/banet_hip.h"

namespace banet {

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kTilePix = 64;      // source pixels per workgroup tile (8x8 patch in dense mode)
constexpr int kBlock = 256;       // 4 waves
constexpr int kNumWaves = kBlock / kWave;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

__device__ __forceinline__ float rfl(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// all-reduce (sum) over the 64 lanes; fixed butterfly order -> deterministic.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// One level of the transposing butterfly: lanes with bit `s` clear keep `a`, lanes with the
// bit set keep `b`; each adds the partner's copy of what it keeps.  After merging with
// s = 32,16,8,4 a lane holds one of 16 values summed over the 16 lanes sharing its bits 5..2.
__device__ __forceinline__ float bfly_merge(float a, float b, int s) {
  const bool hi = (lane_id() & s) != 0;
  const float keep = hi ? b : a;
  const float send = hi ? a : b;
  return keep + __shfl_xor(send, s, 64);
}

// pixel slot (0..15) whose channel-sums end up on this lane after the 4-level merge tree
// built as  level1: s=32 (pixel bit0), level2: s=16 (bit1), level3: s=8 (bit2), level4: s=4 (bit3)
__device__ __forceinline__ int bfly_slot(int lane) {
  return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3);
}

struct Q5 {  // per-pixel channel reductions: M = G^T G (3 unique), g = G^T d (2)
  float m11, m12, m22, g1, g2;
};

__device__ __forceinline__ Q5 q5_merge(const Q5& a, const Q5& b, int s) {
  Q5 r;
  r.m11 = bfly_merge(a.m11, b.m11, s);
  r.m12 = bfly_merge(a.m12, b.m12, s);
  r.m22 = bfly_merge(a.m22, b.m22, s);
  r.g1 = bfly_merge(a.g1, b.g1, s);
  r.g2 = bfly_merge(a.g2, b.g2, s);
  return r;
}

__device__ __forceinline__ Q5 q5_finish(Q5 q) {  // remaining lane bits 1,0
#pragma unroll
  for (int s = 2; s >= 1; s >>= 1) {
    q.m11 += __shfl_xor(q.m11, s, 64);
    q.m12 += __shfl_xor(q.m12, s, 64);
    q.m22 += __shfl_xor(q.m22, s, 64);
    q.g1 += __shfl_xor(q.g1, s, 64);
    q.g2 += __shfl_xor(q.g2, s, 64);
  }
  return q;
}

// reflect-pad neighbour indices of tf.pad(...,'REFLECT') + central difference
// (bundlenet.py:97-99): at the border both neighbours coincide -> zero gradient.
__device__ __forceinline__ int refl_m(int i) { return i == 0 ? 1 : i - 1; }
__device__ __forceinline__ int refl_p(int i, int n) { return i == n - 1 ? n - 2 : i + 1; }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace banet
