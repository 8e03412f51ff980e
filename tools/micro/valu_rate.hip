// What does a wave64 v_fma_f32 cost a SIMD — 2 cycles (32 lanes per clock: the 157 TFLOP/s of the data sheet) or 4 — and
// does v_pk_fma_f32 double the rate?  W waves per SIMD (blocks of 256 * W threads, one block per CU, every CU busy), each
// wave a stream of fmas on CH independent accumulators (CH = 1: one dependent chain, as the pool's last layer has per
// row).  Prints s_memtime ticks (100 MHz here? no: shader clocks via s_memtime) per instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/valu_rate tools/micro/valu_rate.hip && tools/micro/_bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CH, bool PK>
__global__ void k_rate(unsigned long long* out, int reps, float seed) {
  float a[16];
  f32x2 p[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    a[k] = seed + threadIdx.x + k;
    p[k] = f32x2{seed + k, seed - k};
  }
  const float m = 1.0001f + seed;
  const f32x2 m2 = {m, m};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      if (PK) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[u % CH]) : "v"(m2), "v"(p[(u + 1) % 16 == u % CH ? 15 : 15]));
      } else {
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[u % CH]) : "v"(m), "v"(a[15]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a[k] + p[k].x + p[k].y;
  if (s == 12345.678f) out[1] = 1;
  if (threadIdx.x == blockDim.x - 64 && blockIdx.x == 0) out[0] = t1 - t0;   // the YOUNGEST wave of the block (the oldest has issue priority)
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = t1 - t0;
}

template <int CH, bool PK>
static void run(int waves_per_simd, unsigned long long* d) {
  const int reps = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<CH, PK>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d, reps, 0.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_rate<CH, PK>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d, reps, 0.0f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[3] = {0, 0, 0};
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double n = reps * 64.0;
  // instructions one SIMD issued: waves_per_simd * n; in ms milliseconds
  const double ns_per_instr_simd = ms * 1e6 / (waves_per_simd * n);
  printf("%-12s chains %2d  waves/SIMD %d : youngest wave %6.2f ticks/instr, oldest %6.2f; kernel %7.3f ms = %5.3f ns per instruction and SIMD (%4.2f cycles at 2.4 GHz)\n",
         PK ? "v_pk_fma_f32" : "v_fma_f32", CH, waves_per_simd, (double)h[0] / n, (double)h[2] / n, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4);
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  for (int w = 1; w <= 4; ++w) {
    run<1, false>(w, d);
    run<8, false>(w, d);
    run<1, true>(w, d);
    run<8, true>(w, d);
  }
  // s_memtime against the shader clock: a known-cost loop (64-cycle MFMA) would calibrate; here: s_sleep
  return 0;
}
