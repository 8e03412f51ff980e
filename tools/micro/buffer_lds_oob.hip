// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS to the LDS (or leave it alone)?  gemm_v2 wants
// the zero row of a missing lattice neighbour from the buffer's range check instead of a select against a zero buffer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/buffer_lds_oob tools/micro/buffer_lds_oob.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* out, int n_floats) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  float* f = reinterpret_cast<float*>(sm);
  for (int i = threadIdx.x; i < 64 * 4; i += blockDim.x) f[i] = 7.0f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_floats * 4, 0x00020000);
  // even lanes read their 16 bytes, odd lanes an offset far out of range
  const unsigned off = (threadIdx.x & 1) ? 0x80000000u + threadIdx.x * 16 : threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sm, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4; i += blockDim.x) out[i] = f[i];
}
int main() {
  float h[256], *d_src, *d_out;
  for (int i = 0; i < 256; ++i) h[i] = 100.0f + i;
  (void)hipMalloc(&d_src, sizeof(h));
  (void)hipMalloc(&d_out, sizeof(h));
  (void)hipMemcpy(d_src, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 64 * 16, 0, d_src, d_out, 256);
  (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  int zeros = 0, kept = 0, data = 0, other = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 4; ++e) {
      const float v = h[lane * 4 + e];
      if (lane & 1) {
        if (v == 0.0f) ++zeros; else if (v == 7.0f) ++kept; else ++other;
      } else {
        if (v == 100.0f + lane * 4 + e) ++data; else ++other;
      }
    }
  printf("in-range lanes: %d of 128 values correct; out-of-range lanes: %d zeros, %d left alone, %d other; total other %d\n", data, zeros, kept, other - 0, other);
  return 0;
}
