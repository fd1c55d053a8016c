// Probe of gfx950 VALU issue rates as the gather kernels see them: cycles per instruction (s_memtime) of independent and dependent
// chains of v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 / v_add_f32 with a DPP quad_perm / v_cndmask, for ONE wave per
// SIMD and for TWO waves per SIMD (the strip gather's occupancy), plus the return time of 12 ds_read_b128.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned long long tick() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

typedef float v2 __attribute__((ext_vector_type(2)));

// TEST: 0 v_fma independent (8 accumulators), 1 v_fma dependent, 2 v_pk_fma independent, 3 v_pk_fma dependent, 4 v_pk_add indep,
// 5 v_pk_mul indep, 6 v_add_f32_dpp quad_perm indep, 7 v_cndmask indep, 8 mix: 1 v_pk_fma + 2 v_fma, 9: 12 x ds_read_b128 + wait
template <int TEST>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[64 * 64];
  const int lane = threadIdx.x & 63;
  float a[8];
  v2 p[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = (float)(lane + i);
    p[i] = v2{(float)(lane + i), (float)(lane - i)};
  }
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const float m = 1.0000001f, c = 1e-9f;
  const v2 pm = {m, m}, pc = {c, c};
  unsigned long long t0 = tick();
  for (int it = 0; it < iters; ++it) {
    if constexpr (TEST == 0) {
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(m), "v"(c));)
    } else if constexpr (TEST == 1) {
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                         : "+v"(a[0])
                         : "v"(m), "v"(c));)
    } else if constexpr (TEST == 2) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                         "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                         : "v"(pm), "v"(pc));)
    } else if constexpr (TEST == 3) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\t"
                         "v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %1, %2"
                         : "+v"(p[0])
                         : "v"(pm), "v"(pc));)
    } else if constexpr (TEST == 4) {
      REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n\tv_pk_add_f32 %1, %1, %8\n\tv_pk_add_f32 %2, %2, %8\n\tv_pk_add_f32 %3, %3, %8\n\t"
                         "v_pk_add_f32 %4, %4, %8\n\tv_pk_add_f32 %5, %5, %8\n\tv_pk_add_f32 %6, %6, %8\n\tv_pk_add_f32 %7, %7, %8"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                         : "v"(pc));)
    } else if constexpr (TEST == 5) {
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\t"
                         "v_pk_mul_f32 %4, %4, %8\n\tv_pk_mul_f32 %5, %5, %8\n\tv_pk_mul_f32 %6, %6, %8\n\tv_pk_mul_f32 %7, %7, %8"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                         : "v"(pm));)
    } else if constexpr (TEST == 6) {
      REP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
    } else if constexpr (TEST == 7) {
      REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                         "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(m)
                         : "vcc");)
    } else if constexpr (TEST == 8) {   // 4 v_pk_fma + 4 v_fma per group: 8 instructions
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %4, %4, %10, %11\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %5, %5, %10, %11\n\t"
                         "v_pk_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %6, %6, %10, %11\n\tv_pk_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %7, %7, %10, %11"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(pm), "v"(pc), "v"(m), "v"(c));)
    } else {   // 12 ds_read_b128 (conflict-free: lane i reads 16 B at 16 i + 1024 j) then wait; counted as 12 "instructions"
      const float* base = lds + lane * 4;
      float4 r[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) r[j] = *reinterpret_cast<const float4*>(base + 256 * (j & 7) + (j >> 3) * 2048 % 4096);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 12; ++j) a[j & 7] += (r[j].x + r[j].y) + (r[j].z + r[j].w);
    }
  }
  unsigned long long t1 = tick();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int TEST>
static void run(const char* name, int per_iter) {
  float* o;
  unsigned long long* c;
  hipMalloc(&o, 256 * 512 * 4);
  hipMalloc(&c, 256 * 8 * 8);
  const int iters = 200;
  for (int waves = 4; waves <= 8; waves += 4) {            // 4 waves = one per SIMD, 8 = two per SIMD
    for (int grid = 1; grid <= 256; grid *= 256) {          // one CU / the whole chip
      k<TEST><<<grid, 64 * waves>>>(o, c, iters);
      k<TEST><<<grid, 64 * waves>>>(o, c, iters);
      hipDeviceSynchronize();
      unsigned long long h[256 * 8];
      hipMemcpy(h, c, sizeof(unsigned long long) * grid * waves, hipMemcpyDeviceToHost);
      double mean = 0;
      for (int i = 0; i < grid * waves; ++i) mean += (double)h[i];
      mean /= grid * waves;
      printf("%-34s waves/SIMD %d  workgroups %3d : %7.2f cycles per instruction per wave  (%.2f per SIMD)\n", name, waves / 4, grid,
             mean / ((double)iters * per_iter), mean / ((double)iters * per_iter) / (waves / 4));
    }
  }
  hipFree(o);
  hipFree(c);
}

// ---- second table: one-instruction-type streams of 8 x 16 instructions on 8 independent registers ---------------------------------
#define ASM8(I)  asm volatile(I(0) "\n\t" I(1) "\n\t" I(2) "\n\t" I(3) "\n\t" I(4) "\n\t" I(5) "\n\t" I(6) "\n\t" I(7)                                   \
                      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+s"(sg)            \
                      : "v"(m), "s"(mask), "s"(sc)                                                                                         \
                      : "vcc");
#define I_CND64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %9, %10"
#define I_CNDVCC(k) "v_cndmask_b32 %" #k ", %" #k ", %9, vcc"
#define I_CMPCND(k) "v_cmp_lt_f32 vcc, %9, %" #k "\n\tv_cndmask_b32 %" #k ", %" #k ", %9, vcc"
#define I_READLANE(k) "v_readlane_b32 %8, %" #k ", 3"
#define I_DPPMOV(k) "v_mov_b32_dpp %" #k ", %" #k " quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
#define I_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", 3, %9"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %9"
#define I_MINU(k) "v_min_u32 %" #k ", %" #k ", %9"
#define I_SADD(k) "s_add_u32 %8, %8, %11"
#define I_MULS(k) "v_mul_f32 %" #k ", %11, %" #k
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %9"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %9"
#define I_MULF(k) "v_mul_f32 %" #k ", %" #k ", %9"
#define I_SNOP(k) "s_nop 0"
#define I_SMOVM0(k) "s_mov_b32 m0, %11"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 4, 8"
template <int TEST>
__global__ void k2(float* out, unsigned long long* cyc, int iters, unsigned long long mask, float sc) {
  const int lane = threadIdx.x & 63;
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = (float)(lane + i);
  const float m = 1.0000001f;
  unsigned sg = 1;
  asm volatile("s_mov_b64 vcc, %0" ::"s"(mask) : "vcc");
  unsigned long long t0 = tick();
  for (int it = 0; it < iters; ++it) {
    if constexpr (TEST == 0) { REP16(ASM8(I_CND64)) }
    else if constexpr (TEST == 1) { REP16(ASM8(I_CNDVCC)) }
    else if constexpr (TEST == 2) { REP16(ASM8(I_CMPCND)) }
    else if constexpr (TEST == 3) { REP16(ASM8(I_READLANE)) }
    else if constexpr (TEST == 4) { REP16(ASM8(I_DPPMOV)) }
    else if constexpr (TEST == 5) { REP16(ASM8(I_MAD24)) }
    else if constexpr (TEST == 6) { REP16(ASM8(I_LSHLADD)) }
    else if constexpr (TEST == 7) { REP16(ASM8(I_MINU)) }
    else if constexpr (TEST == 8) { REP16(ASM8(I_SADD)) }
    else if constexpr (TEST == 9) { REP16(ASM8(I_MULS)) }
    else if constexpr (TEST == 10) { REP16(ASM8(I_ADDU)) }
    else if constexpr (TEST == 11) { REP16(ASM8(I_MAX)) }
    else if constexpr (TEST == 12) { REP16(ASM8(I_MULF)) }
    else if constexpr (TEST == 13) { REP16(ASM8(I_SNOP)) }
    else if constexpr (TEST == 14) { REP16(ASM8(I_SMOVM0)) }
    else { REP16(ASM8(I_BFE)) }
  }
  unsigned long long t1 = tick();
  float s = (float)sg;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int TEST>
static void run2(const char* name, int per_iter) {
  float* o;
  unsigned long long* c;
  hipMalloc(&o, 256 * 512 * 4);
  hipMalloc(&c, 256 * 8 * 8);
  const int iters = 200;
  for (int waves = 4; waves <= 8; waves += 4) {
    k2<TEST><<<1, 64 * waves>>>(o, c, iters, 0x5555aaaa5555aaaaull, 1.0000001f);
    k2<TEST><<<1, 64 * waves>>>(o, c, iters, 0x5555aaaa5555aaaaull, 1.0000001f);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, c, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < waves; ++i) mean += (double)h[i];
    mean /= waves;
    printf("%-34s waves/SIMD %d : %7.2f cycles per instruction per wave  (%.2f per SIMD)\n", name, waves / 4,
           mean / ((double)iters * per_iter), mean / ((double)iters * per_iter) / (waves / 4));
  }
  hipFree(o);
  hipFree(c);
}

int main() {
  run<0>("v_fma_f32 independent", 128);
  run<1>("v_fma_f32 dependent chain", 128);
  run<2>("v_pk_fma_f32 independent", 128);
  run<3>("v_pk_fma_f32 dependent chain", 128);
  run<4>("v_pk_add_f32 independent", 128);
  run<5>("v_pk_mul_f32 independent", 128);
  run<6>("v_add_f32_dpp quad_perm indep", 128);
  run<7>("v_cndmask_b32 independent", 128);
  run<8>("v_pk_fma + v_fma alternating", 128);
  run<9>("12 x ds_read_b128 + wait", 12);
  run2<0>("v_cndmask_b32_e64 (SGPR-pair mask)", 128);
  run2<1>("v_cndmask_b32 (vcc, set once)", 128);
  run2<2>("v_cmp_lt_f32 vcc + v_cndmask", 256);
  run2<3>("v_readlane_b32", 128);
  run2<4>("v_mov_b32_dpp quad_perm", 128);
  run2<5>("v_mad_u32_u24", 128);
  run2<6>("v_lshl_add_u32", 128);
  run2<7>("v_min_u32", 128);
  run2<8>("s_add_u32 (dependent)", 128);
  run2<9>("v_mul_f32 with an SGPR operand", 128);
  run2<10>("v_add_u32", 128);
  run2<11>("v_max_f32", 128);
  run2<12>("v_mul_f32", 128);
  run2<13>("s_nop 0", 128);
  run2<14>("s_mov_b32 m0", 128);
  run2<15>("v_bfe_u32", 128);
  return 0;
}
