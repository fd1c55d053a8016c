// Probe of the gfx950 cross-lane primitives used by the butterfly: prints, for every lane, which
// source lane each primitive delivers.   hipcc --offload-arch=gfx950 lanes.hip -o lanes && ./lanes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ int dppi(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xF, 0xF, true); }
__global__ void k(int* o) {
  const int l = threadIdx.x;
  const int a = l, b = 100 + l;
  u2 r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
  o[0 * 64 + l] = r[0]; o[1 * 64 + l] = r[1];
  u2 q = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
  o[2 * 64 + l] = q[0]; o[3 * 64 + l] = q[1];
  o[4 * 64 + l] = dppi<0x128>(l);
  o[5 * 64 + l] = dppi<0x141>(l);
  o[6 * 64 + l] = dppi<0x4E>(l);
  o[7 * 64 + l] = dppi<0xB1>(l);
  o[8 * 64 + l] = dppi<0x140>(l);
}
int main() {
  int* d; hipMalloc(&d, 9 * 64 * 4); k<<<1, 64>>>(d); int h[9 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[9] = {"swap32 r0 (a')", "swap32 r1 (b')", "swap16 r0 (a')", "swap16 r1 (b')", "dpp row_ror:8", "dpp half_mirror", "dpp quad[2,3,0,1]", "dpp quad[1,0,3,2]", "dpp row_mirror"};
  for (int i = 0; i < 9; ++i) { printf("%-18s:", names[i]); for (int l = 0; l < 64; ++l) printf(" %d", h[i * 64 + l]); printf("\n"); }
  return 0;
}
