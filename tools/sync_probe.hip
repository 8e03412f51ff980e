// How long does the host wait for a small device result?  (a) kernel + hipMemcpyAsync D2H + hipStreamSynchronize,
// (b) kernel that stores the result and a ticket into mapped pinned host memory, host spins on the ticket,
// (c) as (a) but with a 20 us kernel in front (the frame's real situation: the queue is not empty).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/sync_probe tools/sync_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_result(int* d, int v) { if (threadIdx.x == 0) d[0] = v; }
__global__ void k_result_host(volatile int* h, int v) {
  if (threadIdx.x == 0) {
    h[0] = v;
    __atomic_store_n((int*)h + 16, v, __ATOMIC_RELEASE);   // system scope by default for a host allocation
  }
}
__global__ void k_busy(float* x, int iters) {
  float a = x[threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  x[threadIdx.x] = a;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  int *d, *h, *hd;
  float* x;
  CK(hipMalloc(&d, 64));
  CK(hipMalloc(&x, 4096));
  CK(hipMemset(x, 0, 4096));
  CK(hipHostMalloc(&h, 256, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&hd, h, 0));
  h[16] = 0;
  const int N = 2000;
  for (int busy = 0; busy < 2; ++busy) {
    const int iters = busy ? 20000 : 0;
    double t0 = now_us();
    for (int i = 1; i <= N; ++i) {
      if (busy) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, x, iters);
      hipLaunchKernelGGL(k_result, dim3(1), dim3(64), 0, s, d, i);
      CK(hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      if (h[0] != i) { printf("wrong value\n"); return 1; }
    }
    double ta = (now_us() - t0) / N;
    t0 = now_us();
    for (int i = 1; i <= N; ++i) {
      if (busy) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, x, iters);
      hipLaunchKernelGGL(k_result_host, dim3(1), dim3(64), 0, s, hd, i);
      while (__atomic_load_n(h + 16, __ATOMIC_ACQUIRE) != i) {}
      if (h[0] != i) { printf("wrong value\n"); return 1; }
    }
    double tb = (now_us() - t0) / N;
    t0 = now_us();
    for (int i = 1; i <= N; ++i) {
      if (busy) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, x, iters);
      hipLaunchKernelGGL(k_result_host, dim3(1), dim3(64), 0, s, hd, i + N);
      CK(hipStreamSynchronize(s));
      if (h[0] != i + N) { printf("wrong value\n"); return 1; }
    }
    double tc = (now_us() - t0) / N;
    // the busy kernel alone, for reference
    t0 = now_us();
    for (int i = 1; i <= N; ++i) {
      if (busy) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, x, iters);
      CK(hipStreamSynchronize(s));
    }
    double td = (now_us() - t0) / N;
    printf("%s: memcpy+sync %.1f us | mapped store + spin %.1f us | mapped store + hipStreamSynchronize %.1f us | "
           "front kernel + sync only %.1f us\n", busy ? "busy queue" : "empty queue", ta, tb, tc, td);
  }
  return 0;
}
