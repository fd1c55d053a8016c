// Read-only streaming rate of this part, as a calibration of "the memory system's ceiling" for kernels that mostly read:
// 2048 resident waves (256 CUs x 8), 16-byte loads, U loads in flight per lane, a buffer far larger than the 256 MB
// infinity cache.  hipcc --offload-arch=gfx950 -O3 tools/probe/stream_read.hip -o tools/probe/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ p, size_t n, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;   // keep the loads
}

template <int U, bool NT>
static void run(const f32x4* p, size_t n, float* out, int wgs, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(wgs), dim3(256), 0, 0, p, n, out);
  hipEventRecord(e0, 0);
  const int reps = n * 16 < ((size_t)1 << 30) ? 200 : 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(wgs), dim3(256), 0, 0, p, n, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %d workgroups: %.1f GB/s\n", name, wgs, (double)n * 16 * reps / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  // stream_read [MiB]: the buffer size (default 8 GiB, far beyond the 256 MiB infinity cache; 64 .. 192 = resident in it:
  // does a re-read that hits the infinity cache go faster than one from HBM?)
  const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) << 20 : (size_t)8 << 30, n = bytes / 16;
  printf("buffer %zu MiB\n", bytes >> 20);
  f32x4* p;
  float* out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(p, 0, bytes);
  for (int wgs : {512, 768, 1024, 2048}) {
    run<4, false>(p, n, out, wgs, "16 B loads, 4 in flight");
    run<8, false>(p, n, out, wgs, "16 B loads, 8 in flight");
    run<8, true>(p, n, out, wgs, "16 B nontemporal loads, 8 in flight");
  }
  hipDeviceSynchronize();
  return 0;
}
