// Shared helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tln.h"

#define TLN_WAVE 64

void tln_set_error(const char* fmt, ...);
const tln_options& tln_opt(const tln_options* o);   // *o, or the immutable defaults for NULL (gemm.hip)

#define TLN_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t _e = (call);                                                             \
    if (_e != hipSuccess) {                                                             \
      tln_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return TLN_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

#define TLN_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      tln_set_error(__VA_ARGS__);         \
      return TLN_E_INVALID;               \
    }                                     \
  } while (0)

#define TLN_LAUNCH_CHECK()                                                              \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) {                                                             \
      tln_set_error("%s:%d: launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e));  \
      return TLN_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

static inline int64_t tln_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel, set once per host thread, device and kernel (a driver call
// per launch otherwise: a frame has a hundred launches that need it).  `slot` is a static thread_local of the caller.
struct TlnLdsAttr {
  int device = -1;
  int bytes = 0;
};
static inline hipError_t tln_set_max_lds(TlnLdsAttr& slot, const void* kern, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (slot.device == dev && slot.bytes >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) {
    slot.device = dev;
    slot.bytes = bytes;
  }
  return e;
}

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ int tln_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ double tln_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float tln_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int tln_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Jobs of a batched launch <-> XCDs (grid = [blocks, jobs], blockIdx.y = job: the frames / lattices of lock-stepped
// sequences).  Workgroups are dealt round-robin over the eight XCDs in their linear order (observed; used for speed only,
// nothing depends on it) and every XCD has an L2 of its own, so with the job taken from the LOW bits of the linear block
// index all blocks of a job run on one XCD (8 jobs; a pair with 4, four with 2): what they gather from or scatter into —
// one frame's points, one lattice's values — meets in ONE 4 MB L2 instead of passing through all eight.
__device__ __forceinline__ void tln_xcd_block(int on, int& bx, int& job) {
  bx = (int)blockIdx.x;
  job = (int)blockIdx.y;
  const unsigned ny = gridDim.y;
  if (on && (ny == 8u || ny == 4u || ny == 2u)) {
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
    job = (int)(L & (ny - 1u));
    bx = (int)(L / ny);
  }
}
// (host: env TLN_XCD=0 switches the mapping off everywhere, for measurements)
static inline int tln_xcd_on() {
  static const int on = (getenv("TLN_XCD") != nullptr && atoi(getenv("TLN_XCD")) == 0) ? 0 : 1;
  return on;
}

// ONE arithmetic for the GRU cell wherever it is evaluated (torch.nn.GRUCell, reference lattice_modules.py:62): the fused
// large-lattice cell (gemm_v2.hip epilogue) and the small-lattice gates kernel (fused.hip k_gru_gates) call this, so a
// lattice that crosses the size threshold does not change its gate arithmetic.  Accurate library exp / tanh and a
// correctly rounded division (round 3's fused cell used the hardware exp2 / rcp approximations and tanh by exp); every
// multiply-add is spelled out as one fused operation so that translation units with different -ffp-contract settings and
// different instantiations of one epilogue produce the same bits.
//   pre_r = gi_r + gh_r (+ biases), pre_z likewise, gi_n / gh_n = the two halves of the n gate, h = padded hidden state
__device__ __forceinline__ float tln_sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tln_gru_cell_value(float pre_r, float pre_z, float gi_n, float gh_n, float h) {
#ifdef TLN_GRU_FAST_GATES   // measurement builds only (TLN_EXTRA_FLAGS=-DTLN_GRU_FAST_GATES): round 3's approximations,
                            // kept to put a number on what they contributed (tools/parity64.py)
  const float rr = __frcp_rn(1.0f + __expf(-pre_r));
  const float zz = __frcp_rn(1.0f + __expf(-pre_z));
  const float na = fmaf(rr, gh_n, gi_n);
  const float nn = fmaf(-2.0f, __frcp_rn(1.0f + __expf(2.0f * na)), 1.0f);
  return fmaf(zz, h, __fmul_rn(1.0f - zz, nn));
#endif
  const float r = tln_sigmoid_acc(pre_r);
  const float z = tln_sigmoid_acc(pre_z);
  const float n = tanhf(fmaf(r, gh_n, gi_n));
  return fmaf(z, h, __fmul_rn(1.0f - z, n));
}

// order-preserving map float -> uint32 (larger float => larger uint)
__device__ __forceinline__ uint32_t tln_f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float tln_ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// fixed-point positions for the per-vertex local mean: units of 2^-20, summed in int64 — exact, so
// the mean does not depend on the order in which a vertex's rows are added (oracle/ops.py:distribute does the same)
__device__ __forceinline__ long long tln_fix20(float p) { return __float2ll_rn(p * 1048576.0f); }
__device__ __forceinline__ float tln_unfix20(long long sum, double cnt) {
  return (float)(((double)sum / cnt) * (1.0 / 1048576.0));
}

// the vertex bins of the last distribute (lattice.hip) as the pool (pool.hip) sees them
struct tln_lattice;
struct __attribute__((aligned(16))) TlnBinRec {
  float4 a;   // position, value
  uint4 m;    // barycentric weight (bits), row id (4 * point + simplex vertex), vertex index or -1, 0
};
struct TlnBins {
  const TlnBinRec* rec;    // [rows] the frame's rows grouped by vertex (rows without a vertex in between)
  const int32_t* vstart;   // [V] first bin position of a vertex
  const int32_t* vcnt;     // [V] rows of the frame on a vertex
  const float* mean;       // [V][3] local mean of the frame (valid when subtract)
  const int32_t* ctr;      // device counters: [0] = V, [6] = rows placed in vertex segments
  const float* weights;    // [rows] barycentric weights in row order
  int subtract;
  const int32_t* vstamp;   // partitioned K1: vcnt[v] / vstart[v] / mean[v] belong to this frame iff vstamp[v] == stamp
  int stamp;               // (NULL: every vertex was visited)
};
bool tln_lat_bins(const tln_lattice* l, const float* d_distributed, int64_t rows, TlnBins* out);
const tln_options& tln_lat_options(const tln_lattice* l);   // the handle's kernel-selection options (tln_lattice_set_options)

// ---- key packing / hashing (d = 3) ------------------------------------------------------
#define TLN_KEY_BIAS (1 << 20)
#define TLN_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull

__host__ __device__ __forceinline__ bool tln_key_in_range(int k0, int k1, int k2) {
  const int lim = TLN_KEY_BIAS;
  return k0 >= -lim && k0 < lim && k1 >= -lim && k1 < lim && k2 >= -lim && k2 < lim;
}
__host__ __device__ __forceinline__ uint64_t tln_pack_key(int k0, int k1, int k2) {
  return ((uint64_t)(uint32_t)(k0 + TLN_KEY_BIAS) << 42) | ((uint64_t)(uint32_t)(k1 + TLN_KEY_BIAS) << 21) |
         (uint64_t)(uint32_t)(k2 + TLN_KEY_BIAS);
}
__host__ __device__ __forceinline__ void tln_unpack_key(uint64_t p, int& k0, int& k1, int& k2) {
  k0 = (int)((p >> 42) & 0x1FFFFF) - TLN_KEY_BIAS;
  k1 = (int)((p >> 21) & 0x1FFFFF) - TLN_KEY_BIAS;
  k2 = (int)(p & 0x1FFFFF) - TLN_KEY_BIAS;
}
__host__ __device__ __forceinline__ uint64_t tln_mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}
