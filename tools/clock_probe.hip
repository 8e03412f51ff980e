// diagnostic: shader clock under different load patterns (s_memtime ticks / s_memrealtime @100 MHz)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_probe(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0001f, 0.5f);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
__global__ void k_tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; }
int main() {
  unsigned long long* d; hipMalloc(&d, 64); float* f; hipMalloc(&f, 4096);
  unsigned long long h[3];
  auto probe = [&](const char* what, int blocks, int iters) {
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-40s shader ticks %8llu real(100MHz) %6llu -> %.2f GHz\n", what, h[0], h[1], h[0] / (h[1] * 10.0) );
  };
  probe("cold, 1 block, 20k fma", 1, 20000);
  probe("again", 1, 20000);
  probe("1 block, 2k fma (short kernel)", 1, 2000);
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(64), 0, 0, f);
  probe("after 2000 tiny launches, short", 1, 2000);
  probe("1024 blocks, 200k fma (busy chip)", 1024, 200000);
  probe("right after busy, short", 1, 2000);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(64), 0, 0, f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("1000 dependent tiny launches: %.2f us each\n", ms);
  return 0;
}
