// L1/TA cost of the direct gather-GEMM's A-operand pattern.  Every wave reads 32 rows x 128 B per "chunk" with four
// dwordx4 loads: (a) as the kernel does -- lane = (row, half), instruction q fetches the lane's q-th 16 B piece
// (32 lines touched by every instruction), (b) line-coalesced -- instruction q fetches rows 8q..8q+7 whole
// (8 lines per instruction), (c) dword loads of 2 rows x 128 B (the [K,N] weight pattern), 16 per chunk.
// Rows are scattered over a 2 MB (L2-resident) buffer.   hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_probe(const float* __restrict__ src, const int* __restrict__ rows, int nrows,
                                               int chunks, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < chunks; ++c) {
    const int base = ((wv * chunks + c) * 32) % nrows;
    if (MODE == 0) {
      const int r = rows[base + (lane & 31)];
      const f32x4* p = reinterpret_cast<const f32x4*>(src + (size_t)r * 32 + 16 * (lane >> 5));
#pragma unroll
      for (int q = 0; q < 4; ++q) acc += p[q];
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rows[base + 8 * q + (lane >> 3)];
        acc += *reinterpret_cast<const f32x4*>(src + (size_t)r * 32 + 4 * (lane & 7));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int r = rows[base + 2 * q + (lane >> 5)];
        acc[q & 3] += src[(size_t)r * 32 + (lane & 31)];
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  const int nrows = 16384;  // x 128 B = 2 MB
  std::vector<int> h(nrows);
  for (int i = 0; i < nrows; ++i) h[i] = (int)(((long long)i * 7919) % nrows);
  float *src, *out;
  int* rows;
  CK(hipMalloc(&src, (size_t)nrows * 128));
  CK(hipMemset(src, 0, (size_t)nrows * 128));
  CK(hipMalloc(&rows, nrows * 4));
  CK(hipMemcpy(rows, h.data(), nrows * 4, hipMemcpyHostToDevice));
  const int blocks = 1024, chunks = 64;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[3] = {"(a) lane = (row, half), 4 x dwordx4, 32 lines / instr", "(b) whole rows, 4 x dwordx4, 8 lines / instr",
                          "(c) 16 x dword, 2 lines / instr"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, 0));
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 0, 0, src, rows, nrows, chunks, out);
      if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 0, 0, src, rows, nrows, chunks, out);
      if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(256), 0, 0, src, rows, nrows, chunks, out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double bytes = (double)blocks * 4 * chunks * 4096;
    printf("%-58s %.1f us  %.2f TB/s  %.1f B/clk/CU @2.4GHz\n", names[mode], best * 1e3, bytes / best / 1e9,
           bytes / (best * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
