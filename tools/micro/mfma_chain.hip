// Does v_mfma_f32_32x32x2_f32 accumulate like a chain of fp32 fmas in ascending k?  (tools/micro: measurement only)
// C[m][n] = acc; for k: acc = fma(A[m][k], B[k][n], acc)  vs the MFMA result, bitwise, for K = 2 .. 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int K>
__global__ void k_mfma(const float* A, const float* B, const float* C0, float* C) {
  // A [32][K], B [K][32], C [32][32]
  const int lane = threadIdx.x;
  f16v acc;
  for (int i = 0; i < 16; ++i) {
    const int row = (lane / 32) * 4 + 8 * (i / 4) + (i % 4), col = lane % 32;
    acc[i] = C0[row * 32 + col];
  }
  for (int k = 0; k < K; k += 2) {
    const float a = A[(lane % 32) * K + k + lane / 32];
    const float b = B[(k + lane / 32) * 32 + lane % 32];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (lane / 32) * 4 + 8 * (i / 4) + (i % 4), col = lane % 32;
    C[row * 32 + col] = acc[i];
  }
}

template <int K>
int run(unsigned seed, float scale) {
  std::vector<float> A(32 * K), B(K * 32), C0(1024), C(1024), R(1024);
  srand(seed);
  auto rnd = [&]() { return scale * ((float)rand() / RAND_MAX * 2.0f - 1.0f); };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  for (auto& v : C0) v = rnd();
  float *dA, *dB, *dC0, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC0, 4096); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC0, C0.data(), 4096, hipMemcpyHostToDevice);
  k_mfma<K><<<1, 64>>>(dA, dB, dC0, dC);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  int bad_chain = 0, bad_pair = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      float acc = C0[m * 32 + n], acc2 = acc;
      for (int k = 0; k < K; ++k) acc = fmaf(A[m * K + k], B[k * 32 + n], acc);
      for (int k = 0; k < K; k += 2) {   // alternative: exact pair sum, one rounding
        double s = (double)A[m * K + k] * B[k * 32 + n] + (double)A[m * K + k + 1] * B[(k + 1) * 32 + n] + (double)acc2;
        acc2 = (float)s;
      }
      if (memcmp(&acc, &C[m * 32 + n], 4)) ++bad_chain;
      if (memcmp(&acc2, &C[m * 32 + n], 4)) ++bad_pair;
    }
  printf("K=%2d scale=%g: differs from the fma chain in %d / 1024, from pair-sum in %d / 1024\n", K, scale, bad_chain, bad_pair);
  hipFree(dA); hipFree(dB); hipFree(dC0); hipFree(dC);
  return bad_chain;
}

int main() {
  int bad = 0;
  for (unsigned s = 1; s <= 3; ++s) {
    bad += run<2>(s, 1.0f);
    bad += run<16>(s, 1.0f);
    bad += run<32>(s, 3.0f);
  }
  printf(bad ? "NOT a plain fma chain\n" : "bitwise equal to the ascending fma chain\n");
  return 0;
}
