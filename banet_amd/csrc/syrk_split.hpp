// bf16x3 operand split shared by the bf16x6 SYRK kernels (syrk.hip, syrk_wide.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace banet {

// exact 3-way bf16 split of 8 fp32 values, two at a time: v_cvt_pk_bf16_f32 (round to nearest even) gives the packed
// MFMA operand dword directly; v - hi and (v - hi) - mid are exact in fp32, so hi + mid + lo = v up to 2^-25 |v|.
// 9 VALU instructions per two values (cvt_pk, shift, and, pk_add, ... ) against ~19 for mask-and-subtract per value.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8_bf16x3(const float (&x)[8], u32x4_t (&out)[3]) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    f32x2_t v = {x[2 * d], x[2 * d + 1]};          // k = 2d (low half), 2d + 1 (high half)
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t hf = {__uint_as_float(hp << 16), __uint_as_float(hp & 0xffff0000u)};
    const f32x2_t r1 = v - hf;
    const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
    const f32x2_t mf = {__uint_as_float(mp << 16), __uint_as_float(mp & 0xffff0000u)};
    const f32x2_t r2 = r1 - mf;
    out[0][d] = hp;
    out[1][d] = mp;
    out[2][d] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
  }
}

// two-piece fp16 split of 8 fp32 values, two at a time: hi = fp16(v) (v_cvt_pk_f16_f32, round to nearest even), lo = fp16(v - hi)
// with v - hi exact in fp32: hi + lo = v up to 2^-22 |v| while lo is a normal fp16 number (|v| >= 2^-3), an ABSOLUTE 2^-25
// below that (subnormal lo).  The caller scales v into fp16's range first (|v| < 2^14).  5 VALU instructions per two values.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8_f16x2(const float (&x)[8], u32x4_t (&out)[2]) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const f32x2_t v = {x[2 * d], x[2 * d + 1]};
    const f16x2_t hp = __builtin_convertvector(v, f16x2_t);
    const f32x2_t r1 = v - __builtin_convertvector(hp, f32x2_t);
    out[0][d] = __builtin_bit_cast(unsigned, hp);
    out[1][d] = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, f16x2_t));
  }
}

}  // namespace banet
