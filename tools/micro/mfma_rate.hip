// How many cycles does one v_mfma_f32_32x32x2_f32 hold a SIMD's matrix pipe, and what does a wave get to issue in its
// shadow?  NV independent vector instructions and ND LDS reads behind every MFMA (a scheduling fence per MFMA keeps the
// interleave), W waves per SIMD, every CU busy.  Prints s_memtime ticks per MFMA (64 = the 157 TFLOP/s of the data sheet).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/mfma_rate tools/micro/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int ND>
__global__ void __launch_bounds__(512) k_rate(unsigned long long* out, int reps, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
  f32x16 acc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = seed + threadIdx.x, y = seed * 0.5f;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = x + k;
  float d[4] = {0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      acc[u % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x + d[u & 3], y, acc[u % 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k & 7] = fmaf(v[k & 7], 1.0001f, 0.5f);     // independent chains
#pragma unroll
      for (int k = 0; k < ND; ++k) d[(u + k) & 3] = lds[(lane + 64 * ((u + k + i) & 31))];   // consumed 2+ MFMAs later
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
#pragma unroll
  for (int a = 0; a < 3; ++a) s += acc[a][0] + acc[a][7];
  if (s == 12345.678f) out[1] = 1;
  if (threadIdx.x == blockDim.x - 64 && blockIdx.x == 0) out[0] = t1 - t0;    // the YOUNGEST wave of the block
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = t1 - t0;                  // the oldest
}
// G MFMAs in a row, then their G x NV vector instructions in one burst: does the cost of a burst amortise?
template <int NV, int G>
__global__ void __launch_bounds__(512) k_clump(unsigned long long* out, int reps, float seed) {
  f32x16 acc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = seed + threadIdx.x, y = seed * 0.5f;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = x + k;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u0 = 0; u0 < 12; u0 += G) {
#pragma unroll
      for (int u = u0; u < u0 + G; ++u) acc[u % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[u % 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NV * G; ++k) v[k & 7] = fmaf(v[k & 7], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
#pragma unroll
  for (int a = 0; a < 3; ++a) s += acc[a][0] + acc[a][7];
  if (s == 12345.678f) out[1] = 1;
  if (threadIdx.x == blockDim.x - 64 && blockIdx.x == 0) out[0] = t1 - t0;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = t1 - t0;
}
template <int NV, int G>
void run_clump(int waves_per_simd, unsigned long long* d) {
  const int reps = 1000;
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((k_clump<NV, G>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d, reps, 1.0f);
  unsigned long long h[3] = {0, 0, 0};
  (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("%d MFMAs, then their %d x %d vector instructions in one burst, %d wave(s)/SIMD: %.1f ticks per MFMA of the SIMD (oldest %.1f)\n", G, G, NV,
         waves_per_simd, (double)h[0] / (reps * 12.0) / waves_per_simd, (double)h[2] / (reps * 12.0) / waves_per_simd);
}
// ND LDS reads (B128: 16 bytes per lane, else 4) behind every MFMA, fixed addresses, nothing but the reads: what does
// an LDS read cost the matrix pipe?
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ND, bool B128>
__global__ void __launch_bounds__(512) k_lds(unsigned long long* out, int reps, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed * i;
  f32x16 acc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = seed + threadIdx.x, y = seed * 0.5f;
  const unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * (B128 ? 16 : 4);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      acc[u % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[u % 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        if (B128) {
          f32x4 d;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(1024 * ((7 * k) % 8)));
          asm volatile("" ::"v"(d));
        } else {
          float d;
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(256 * ((5 * k) % 16)));
          asm volatile("" ::"v"(d));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) s += acc[a][0] + acc[a][7];
  if (s == 12345.678f) out[1] = 1;
  if (threadIdx.x == blockDim.x - 64 && blockIdx.x == 0) out[0] = t1 - t0;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = t1 - t0;
}
template <int ND, bool B128>
void run_lds(int waves_per_simd, unsigned long long* d) {
  const int reps = 1000;
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((k_lds<ND, B128>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d, reps, 1.0f);
  unsigned long long h[3] = {0, 0, 0};
  (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("%d ds_read_%s per MFMA, %d wave(s)/SIMD: %.1f ticks per MFMA of the SIMD (oldest %.1f)\n", ND, B128 ? "b128" : "b32", waves_per_simd,
         (double)h[0] / (reps * 12.0) / waves_per_simd, (double)h[2] / (reps * 12.0) / waves_per_simd);
}
template <int NV, int ND>
void run(int waves_per_simd, unsigned long long* d) {
  const int reps = 1000;
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((k_rate<NV, ND>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d, reps, 1.0f);
  unsigned long long h[3] = {0, 0, 0};
  (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("%2d vector + %d LDS reads per MFMA, %d wave(s)/SIMD: %.1f ticks per MFMA of the SIMD (youngest wave; oldest %.1f)\n", NV, ND,
         waves_per_simd, (double)h[0] / (reps * 12.0) / waves_per_simd, (double)h[2] / (reps * 12.0) / waves_per_simd);
}
int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, 64);
  (void)hipMemset(d, 0, 64);
  run<0, 0>(1, d); run<0, 0>(2, d);
  run<2, 0>(1, d); run<2, 0>(2, d);
  run<4, 0>(1, d); run<4, 0>(2, d);
  run<8, 0>(1, d); run<8, 0>(2, d);
  run<12, 0>(1, d); run<12, 0>(2, d);
  run<0, 1>(1, d); run<0, 1>(2, d);
  run<0, 2>(1, d); run<0, 2>(2, d);
  run<4, 1>(1, d); run<4, 1>(2, d);
  run<4, 2>(1, d); run<4, 2>(2, d);
  run<8, 2>(1, d); run<8, 2>(2, d);
  run_lds<1, false>(1, d); run_lds<1, false>(2, d);
  run_lds<2, false>(1, d); run_lds<2, false>(2, d);
  run_lds<4, false>(1, d); run_lds<4, false>(2, d);
  run_lds<1, true>(1, d); run_lds<1, true>(2, d);
  run_lds<2, true>(1, d); run_lds<2, true>(2, d);
  run_clump<4, 1>(1, d); run_clump<4, 1>(2, d);
  run_clump<4, 3>(1, d); run_clump<4, 3>(2, d);
  run_clump<4, 6>(1, d); run_clump<4, 6>(2, d);
  run_clump<4, 12>(1, d); run_clump<4, 12>(2, d);
  run_clump<2, 12>(1, d); run_clump<2, 12>(2, d);
  run_clump<1, 12>(1, d); run_clump<1, 12>(2, d);
  return 0;
}
