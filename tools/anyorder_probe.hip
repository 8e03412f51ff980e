// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on this GPU?  (hip_ext.h says the flag is not
// supported on GFX9xx for the module-launch entry point.)  Two 1-block kernels of ~50 us each, 50 pairs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_busy(float* x, int iters) {
  float a = x[threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  x[threadIdx.x] = a;
}
int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  float *x, *y;
  CK(hipMalloc(&x, 4096));
  CK(hipMalloc(&y, 4096));
  CK(hipMemset(x, 0, 4096));
  CK(hipMemset(y, 0, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 30000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 50; ++i) {
        hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, x, iters);
        if (mode == 0) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, y, iters);
        else hipExtLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, y, iters);
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%s: %.1f us per pair\n", mode ? "second kernel with hipExtAnyOrderLaunch" : "both in order", ms * 1e3 / 50);
    }
  }
  return 0;
}
