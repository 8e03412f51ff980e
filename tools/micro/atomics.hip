// Rates of the scattered primitives K1 is made of (tools/micro: measurement only): random returning / non-returning
// u32 atomic adds, random 16-byte loads, random 20-byte row stores; 480k operations over a table of 128k entries.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_add_ret(unsigned* t, const int* idx, int n, int* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int)atomicAdd(&t[idx[i]], 1u);
}
__global__ void k_add_noret(unsigned* t, const int* idx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&t[idx[i]], 1u);
}
__global__ void k_add_wg(unsigned* t, const int* idx, int n, int* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int)__hip_atomic_fetch_add(&t[idx[i]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_load16(const uint4* t, const int* idx, int n, int* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint4 v = t[idx[i]]; out[i] = v.x + v.y + v.z + v.w; }
}
__global__ void k_store20(float* t, const int* idx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float* r = t + (size_t)idx[i] * 5; r[0] = i; r[1] = 1; r[2] = 2; r[3] = 3; r[4] = 4; }
}
__global__ void k_store32(float4* t, const int* idx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float4* r = t + (size_t)idx[i] * 2; r[0] = make_float4(i, 1, 2, 3); r[1] = make_float4(4, 5, 6, 7); }
}
__global__ void k_lds_hist(const int* idx, int n, unsigned* out) {   // LDS atomics, 2048 bins per block
  __shared__ unsigned h[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) h[i] = 0;
  __syncthreads();
  int i = blockIdx.x * blockDim.x * 8 + threadIdx.x;
  unsigned acc = 0;
  for (int k = 0; k < 8; ++k, i += blockDim.x) if (i < n) acc += atomicAdd(&h[idx[i] & 2047], 1u);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc + h[5];
}
int main() {
  const int n = 480000, T = 1 << 17;
  std::vector<int> idx(n), perm(n);
  srand(1);
  for (int i = 0; i < n; ++i) idx[i] = rand() % 30000 * 4 % T;   // ~30k distinct targets, 16 rows each
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int i = n - 1; i > 0; --i) { int j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
  int *dIdx, *dPerm, *dOut; unsigned* dT; uint4* dT16; float* dRows;
  hipMalloc(&dIdx, n * 4); hipMalloc(&dPerm, n * 4); hipMalloc(&dOut, n * 4); hipMalloc(&dT, T * 4);
  hipMalloc(&dT16, T * 16); hipMalloc(&dRows, (size_t)n * 32);
  hipMemcpy(dIdx, idx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dPerm, perm.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(dT, 0, T * 4); hipMemset(dT16, 0, T * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int B = 256, G = (n + B - 1) / B;
  auto time = [&](const char* name, auto f) {
    for (int w = 0; w < 3; ++w) f();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %7.2f us  (%.1f G ops/s)\n", name, ms * 50.0f, n / (ms * 50.0f) / 1e3);
  };
  time("atomicAdd returning, agent", [&] { k_add_ret<<<G, B>>>(dT, dIdx, n, dOut); });
  time("atomicAdd no return, agent", [&] { k_add_noret<<<G, B>>>(dT, dIdx, n); });
  time("atomicAdd returning, workgroup", [&] { k_add_wg<<<G, B>>>(dT, dIdx, n, dOut); });
  time("random 16-B loads (2 MB table)", [&] { k_load16<<<G, B>>>(dT16, dIdx, n, dOut); });
  time("random 20-B row stores", [&] { k_store20<<<G, B>>>(dRows, dPerm, n); });
  time("random 32-B row stores", [&] { k_store32<<<G, B>>>((float4*)dRows, dPerm, n); });
  time("LDS atomics 8/thread", [&] { k_lds_hist<<<(n + B * 8 - 1) / (B * 8), B>>>(dIdx, n, (unsigned*)dOut); });
  return 0;
}
