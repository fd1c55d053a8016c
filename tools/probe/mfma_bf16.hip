// Probe: operand layout of v_mfma_f32_16x16x32_bf16 on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_bf16.hip -o /tmp/mfma_bf16 && /tmp/mfma_bf16
// Hypothesis: A[i][k] lives in lane i + 16 (k / 8), element k % 8; B[k][j] in lane j + 16 (k / 8), element k % 8;
// C[i][j] in lane j + 16 (i / 4), register i % 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* C) {   // A [16][32], B [32][16], C [16][16] row-major
  const int lane = threadIdx.x, r = lane & 15, kg = lane >> 4;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)A[r * 32 + 8 * kg + e];
    b[e] = (__bf16)B[(8 * kg + e) * 16 + r];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) C[(4 * kg + i) * 16 + r] = c[i];
}

int main() {
  float hA[16 * 32], hB[32 * 16], hC[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) hA[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k * 5 + j * 2) % 13 - 6);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (fabsf(hC[i] - ref[i]) > 1e-3f) ++bad;
  printf("mfma_f32_16x16x32_bf16 layout hypothesis: %d mismatches of 256\n", bad);
  return bad != 0;
}
