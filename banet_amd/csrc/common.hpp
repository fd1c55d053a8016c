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

// ---- cross-lane primitives (wave64, gfx950) ------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {  // CTRL: LLVM dpp_ctrl encoding
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppRor8 = 0x128;        // lane i <- lane i^8   (rotate by 8 inside a 16-lane row)
constexpr int kDppHalfMirror = 0x141;  // lane i <- lane i^7   (mirror inside each 8-lane half row)
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// sum over the 16 lanes of a DPP row; every lane of the row gets the total (fixed order)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<kDppRor8>(v);
  v += dpp_mov<kDppHalfMirror>(v);
  v += dpp_mov<kDppXor2>(v);
  v += dpp_mov<kDppXor1>(v);
  return v;
}

// One level of the transposing butterfly: lanes with bit `s` clear keep `a`, lanes with the bit
// set keep `b`; each adds its partner's copy of what it keeps.  Partner = a lane that differs in
// bit `s` and agrees in all higher bits (lower bits may differ: s=4 pairs i with i^7).  After the
// levels s = 32,16,8,4(,2,1) a lane holds one value summed over all lanes that share its higher
// bits.  s=32/16 use the gfx950 half/row swaps, s<=8 DPP -- no LDS crossbar (ds_bpermute) traffic.
__device__ __forceinline__ float bfly_merge(float a, float b, int s) {
  // NOTE: the __builtin_amdgcn_permlane{16,32}_swap builtins are miscompiled by hipcc 7.2 when
  // both results are consumed as floats (r[0] + r[1] becomes r[0] + r[0]; tools/probe/bfly.hip),
  // hence inline asm.  The leading s_nop 1 covers the VALU-write -> permlane-read hazard
  // (2 wait states) that the compiler cannot see inside an asm statement.
  if (s == 32) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;  // lanes <32: a_lo + a_hi ; lanes >=32: b_lo + b_hi
  }
  if (s == 16) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;  // even rows: a_r + a_{r+1} ; odd rows: b_{r-1} + b_r
  }
  const bool hi = (lane_id() & s) != 0;
  const float keep = hi ? b : a;
  const float send = hi ? a : b;
  if (s == 8) return keep + dpp_mov<kDppRor8>(send);
  if (s == 4) return keep + dpp_mov<kDppHalfMirror>(send);
  if (s == 2) return keep + dpp_mov<kDppXor2>(send);
  return keep + dpp_mov<kDppXor1>(send);
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
  q.m11 += dpp_mov<kDppXor2>(q.m11);
  q.m12 += dpp_mov<kDppXor2>(q.m12);
  q.m22 += dpp_mov<kDppXor2>(q.m22);
  q.g1 += dpp_mov<kDppXor2>(q.g1);
  q.g2 += dpp_mov<kDppXor2>(q.g2);
  q.m11 += dpp_mov<kDppXor1>(q.m11);
  q.m12 += dpp_mov<kDppXor1>(q.m12);
  q.m22 += dpp_mov<kDppXor1>(q.m22);
  q.g1 += dpp_mov<kDppXor1>(q.g1);
  q.g2 += dpp_mov<kDppXor1>(q.g2);
  return q;
}

// reflect-pad neighbour indices of tf.pad(...,'REFLECT') + central difference
// (bundlenet.py:97-99): at the border both neighbours coincide -> zero gradient.
__device__ __forceinline__ int refl_m(int i) { return i == 0 ? 1 : i - 1; }
__device__ __forceinline__ int refl_p(int i, int n) { return i == n - 1 ? n - 2 : i + 1; }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace banet
