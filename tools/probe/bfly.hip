// Unit probe of the transposing butterfly in common.hpp (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../banet_amd/csrc/common.hpp"
using namespace banet;
__global__ void k(float* o) {
  const int l = threadIdx.x;
  // level-by-level check: value v_j on lane l is (j+1)*1000 + l ; expected sum over lanes = 64*(j+1)*1000 + 2016
  float a = 1000.f + l, b = 2000.f + l;
  o[0 * 64 + l] = bfly_merge(a, b, 32);
  o[1 * 64 + l] = bfly_merge(a, b, 16);
  o[2 * 64 + l] = bfly_merge(a, b, 8);
  o[3 * 64 + l] = bfly_merge(a, b, 4);
  o[4 * 64 + l] = bfly_merge(a, b, 2);
  o[5 * 64 + l] = bfly_merge(a, b, 1);
  // full 4-level tree over 16 values + finish
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = (j + 1) * 1000.f + l;
  float m1[8], m2[4], m3[2];
  for (int j = 0; j < 8; ++j) m1[j] = bfly_merge(v[2 * j], v[2 * j + 1], 32);
  for (int j = 0; j < 4; ++j) m2[j] = bfly_merge(m1[2 * j], m1[2 * j + 1], 16);
  for (int j = 0; j < 2; ++j) m3[j] = bfly_merge(m2[2 * j], m2[2 * j + 1], 8);
  float q = bfly_merge(m3[0], m3[1], 4);
  q += dpp_mov<kDppXor2>(q);
  q += dpp_mov<kDppXor1>(q);
  o[6 * 64 + l] = q;
  o[7 * 64 + l] = (float)bfly_slot(l);
}
int main() {
  float* d; (void)hipMalloc(&d, 8 * 64 * 4); k<<<1, 64>>>(d); float h[8 * 64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[8] = {"merge32", "merge16", "merge8", "merge4", "merge2", "merge1", "tree16", "slot"};
  for (int i = 0; i < 8; ++i) { printf("%-8s:", nm[i]); for (int l = 0; l < 64; ++l) printf(" %.0f", h[i * 64 + l]); printf("\n"); }
  int bad = 0;
  for (int l = 0; l < 64; ++l) { int j = (int)h[7 * 64 + l]; float e = 64.f * (j + 1) * 1000.f + 2016.f; if (h[6 * 64 + l] != e) ++bad; }
  printf("tree16 mismatches: %d\n", bad);
  return 0;
}
