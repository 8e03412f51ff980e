// Shared by pool.hip and legacy.hip: the PointNet MLP's dense layer with its pinned summation order.
#pragma once
#include "common.h"

struct MlpParams {
  const float* w[3];
  const float* b[3];
};

// one dense layer, weights broadcast from LDS (all lanes read the same address)
template <int CIN, int COUT, bool RELU>
__device__ __forceinline__ void dense(const float* __restrict__ w, const float* __restrict__ b, const float (&in)[CIN],
                                      float (&out)[COUT]) {
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    float acc = b[o];
#pragma unroll
    for (int i = 0; i < CIN; ++i) acc = fmaf(w[o * CIN + i], in[i], acc);
    out[o] = RELU ? fmaxf(acc, 0.0f) : acc;
  }
}

// legacy.hip: the matrix-core variant of the bins pool (test / measurement switch tln_pool_config(1); cin = 3 or 4)
int tln_pool_bins_mfma_launch(int cin, const TlnBins& bn, int64_t rows, const float* const* w, const float* const* b,
                              int min_points, unsigned long long* packed, float* d_out, int32_t* d_argrow, hipStream_t s);
