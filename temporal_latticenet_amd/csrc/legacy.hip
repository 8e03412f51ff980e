// legacy.hip — kernels that are NOT on the product's path: kept as checked statements of what was measured, reachable
// only through a test / measurement switch, and compiled apart from the kernels the frame program launches.
//
//   k_pool_bins_mfma   (tln_pool_config(1) / TLN_POOL_MFMA=1)  the 16-32-64 PointNet MLP with its two wide layers on the
//                      matrix cores.  Bit-identical to k_pool_bins (tests/test_gpu_ops.py::
//                      test_pool_on_the_matrix_cores_is_bitwise_the_fma_chain) and no faster: 67.8 vs 66.5 us per frame
//                       — the walk over the rows (compare + selects per row and channel), not the products,
//                      sets the pace.
//
// What is NOT here although it looks old: the per-row-atomic K1 kernels (k_distribute_insert, k_bins_*: val_dim > 1,
// frames beyond 4M rows, the fallback when a bucket's LDS table overflows), the CSR pool (k_pool_chunks: rows edited by
// the caller between distribute and pool) and the tiled k_gather_gemm (channel counts that are no multiple of 32,
// unaligned rows, N < 32) are live fallbacks of the operator API and stay with their families.
#include "pool_common.h"

// ---------------------------------------------------------------------------------------
// The 16-32-64 MLP of k_pool_bins with its two wide layers on the matrix cores.  v_mfma_f32_32x32x2_f32 accumulates
// like a chain of fp32 fmas in ascending k (tools/micro/mfma_chain.hip: bitwise equal to fmaf over k = 0..K-1 for every
// element), so C = bias, then K/2 MFMAs over the k pairs (0,1), (2,3), .. reproduce the specified summation order
//  bit for bit — the arg-max row, which decides a whole output element, does not move.
//   layer 1 (4 -> 16)   lane = row, VALU, as before (64 fmas)
//   layer 2 (16 -> 32)  transposed product  D[o][row] = b2[o] + sum_k W2[o][k] h1[row][k]:  A = W2 (8 registers, loaded
//                       once per wave), B = h1 of a 32-row tile.  The rows of the wave's 64-row chunk are two tiles; one
//                       v_permlane32_swap per k pair turns the per-lane registers (h1[2j], h1[2j+1]) into the B operands
//                       of BOTH tiles (lanes 0-31 <- k even, lanes 32-63 <- k odd).
//   layer 3 (32 -> 64)  D[row][o] = b3[o] + sum_k h2[row][k] W3[o][k]:  A = h2 of the tile — the accumulator of layer 2
//                       holds, per lane, channels {8g + 4h + j} of row lane%32 (h = lane/32): one swap of registers
//                       (i, i+1) of the lower half's channels gives the A operands of k pairs (2j, 2j+1) and (2j+4, 2j+5);
//                       B = W3 (32 registers).  The 32 x 64 result goes to the wave's LDS tile.
//   max / arg-max       lane = output channel walks the 64 rows of the tile exactly as k_pool_bins does.
// A wave strides over chunks, so the weight registers are loaded once for several of them.
// ---------------------------------------------------------------------------------------
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void swap32(float x, float y, float& lo, float& hi) {
  // lo = {x[0:31], y[0:31]},  hi = {x[32:63], y[32:63]}
  const u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  lo = __uint_as_float(r.x);
  hi = __uint_as_float(r.y);
}

// what the walk over a chunk's rows needs besides the tile: per-lane notes of row `lane` and the wave-uniform run masks
struct PoolWalk {
  int v, row;              // vertex (0 for rows without one) and row id of row `lane`
  float w;                 // its barycentric weight
  unsigned long long startmask, endmask, wholemask, maskedmask;   // bit j: row j starts / ends a run of equal vertices, ...
};

template <int CIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_pool_bins_mfma(const TlnBins bn, int64_t rows, int min_points,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ w3, const float* __restrict__ b3,
                                                        unsigned long long* __restrict__ packed, float* __restrict__ out,
                                                        int32_t* __restrict__ argrow) {
  constexpr int H1 = 16, H2 = 32, COUT = 64;
  constexpr int TS = COUT + 1;   // tile row stride in floats
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l32 = lane & 31;
  float* tile = smem + wid * (64 * TS);

  // operands that stay in registers for every chunk of this wave
  float a2[H1 / 2];
#pragma unroll
  for (int j = 0; j < H1 / 2; ++j) a2[j] = w2[l32 * H1 + 2 * j + half];
  float bw3[2][H2 / 2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < H2 / 2; ++j) bw3[nt][j] = w3[(l32 + 32 * nt) * H2 + 2 * j + half];
  float bias2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bias2[i] = b2[8 * (i / 4) + 4 * half + (i % 4)];
  const float bias3[2] = {b3[l32], b3[l32 + 32]};
  const int nv = bn.ctr[0];
  const float w0 = bn.weights[0];   // lm:514: an arg-max row id > V reads the barycentric weight of row 0
  const int64_t chunks = (rows + 63) >> 6;

  // software pipeline over the wave's chunks: the bin rows of chunk k+1 are requested before chunk k is computed, the
  // per-vertex values they point to (segment, mean) while chunk k's products run
  struct RowIn {
    float4 q;
    int v, row;
    float w;
  };
  struct VtxIn {
    float mx, my, mz;
    int st, c;
  };
  auto load_rows = [&](int64_t ch, RowIn& r) {
    const int64_t at = ch * 64 + lane;
    r.v = -1;
    r.row = 0;
    r.w = 0.0f;
    r.q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < chunks && at < rows) {
      r.q = bn.rec[at].a;
      const uint4 meta = bn.rec[at].m;
      r.v = (int)meta.z;
      r.row = (int)meta.y;
      r.w = __uint_as_float(meta.x);
    }
  };
  auto load_vtx = [&](const RowIn& r, VtxIn& g) {
    g.mx = g.my = g.mz = 0.0f;
    g.st = -1;
    g.c = 0;
    if (r.v >= 0) {
      if (bn.subtract) {
        g.mx = bn.mean[3 * r.v];
        g.my = bn.mean[3 * r.v + 1];
        g.mz = bn.mean[3 * r.v + 2];
      }
      g.st = bn.vstart[r.v];
      g.c = bn.vcnt[r.v];
    }
  };

  // ---- the walk.  lane = output channel: running (max, arg-max row, its barycentric weight) per vertex over the rows
  // of the tile; the vertex, row id and weight of row j by v_readlane from the lane that loaded it; run starts and ends
  // from the masks: straight-line code except for the (wave-uniform) branch at the end of a run.
  const int c = lane;
  float best = 0.0f, bwt = 0.0f;
  int brow = 0;
  auto walk_row = [&](const PoolWalk& pw, int j, float val) {
    const bool start = (pw.startmask >> j) & 1ull;                  // wave-uniform
    const int rj = __builtin_amdgcn_readlane(pw.row, j);
    const float wj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(pw.w), j));
    const bool take = start || val > best || (val == best && rj < brow);   // ties: the smallest row id
    best = take ? val : best;
    brow = take ? rj : brow;
    bwt = take ? wj : bwt;
    if ((pw.endmask >> j) & 1ull) {   // wave-uniform: the run of row j's vertex ends here
      const int v = __builtin_amdgcn_readlane(pw.v, j);
      if ((pw.wholemask >> j) & 1ull) {
        const bool masked = (pw.maskedmask >> j) & 1ull;
        const float bary = brow > nv ? w0 : bwt;   // lm:514
        out[(int64_t)v * (2 * COUT) + c] = masked ? 0.0f : best;
        out[(int64_t)v * (2 * COUT) + COUT + c] = masked ? 0.0f : bary;
        if (argrow) argrow[(int64_t)v * COUT + c] = masked ? -1 : brow;
      } else {
        const unsigned long long p = ((unsigned long long)tln_f2ord(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)brow);
        atomicMax(&packed[(int64_t)v * COUT + c], p);
      }
    }
  };

  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t chunk = (int64_t)blockIdx.x * 4 + wid;
  RowIn rin, rnext;
  VtxIn vin, vnext;
  load_rows(chunk, rin);
  load_vtx(rin, vin);
  for (; chunk < chunks; chunk += stride) {
    const int64_t j0 = chunk * 64;
    const int cnt = (int)((rows - j0) < 64 ? (rows - j0) : 64);
    load_rows(chunk + stride, rnext);
    // ---- lane = row: mean subtraction, layer 1
    float h1[H1];
#pragma unroll
    for (int k = 0; k < H1; ++k) h1[k] = 0.0f;
    PoolWalk cur;
    cur.v = 0;   // rows without a vertex fold into vertex 0 with their raw position (lm:480)
    cur.row = 0;
    cur.w = 0.0f;
    bool r_whole = false, r_masked = false;
    if (lane < cnt) {
      float4 q = rin.q;
      if (rin.v >= 0) {
        cur.v = rin.v;
        q.x -= vin.mx;
        q.y -= vin.my;
        q.z -= vin.mz;
        // every row of a vertex gets the same two answers: do all its rows lie inside this chunk (vertex 0 also takes
        // the rows without a vertex: never), is it below min_points
        r_whole = rin.v != 0 && (int64_t)vin.st >= j0 && (int64_t)vin.st + vin.c <= j0 + cnt;
        r_masked = vin.c < min_points;
      }
      cur.row = rin.row;
      cur.w = rin.w;
      const float xin[4] = {q.x, q.y, q.z, q.w};
      float x[CIN];
#pragma unroll
      for (int k = 0; k < CIN; ++k) x[k] = xin[k];
      dense<CIN, H1, true>(w1, b1, x, h1);
    }
    // wave-uniform masks over the 64 rows: first / last row of a run of equal vertices, whole, masked
    const int v_prev = __shfl_up(cur.v, 1, 64);
    const unsigned long long live = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
    cur.startmask = __ballot(lane == 0 || cur.v != v_prev) & live;
    cur.endmask = ((cur.startmask >> 1) | (1ull << (cnt - 1))) & live;
    cur.wholemask = __ballot(r_whole);
    cur.maskedmask = __ballot(r_masked);
    // ---- layer 2 on both 32-row tiles
    f16v acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc0[i] = bias2[i];
      acc1[i] = bias2[i];
    }
#pragma unroll
    for (int j = 0; j < H1 / 2; ++j) {
      float t0, t1;
      swap32(h1[2 * j], h1[2 * j + 1], t0, t1);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], t0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], t1, acc1, 0, 0, 0);
    }
    // ---- layer 3, tile by tile: 2 x 16 MFMAs, the 32 x 64 result to the wave's LDS tile
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f16v h2 = t ? acc1 : acc0;
#pragma unroll
      for (int i = 0; i < 16; ++i) h2[i] = fmaxf(h2[i], 0.0f);
      float aop[H2 / 2];
#pragma unroll
      for (int jj = 0; jj < H2 / 2; ++jj) {
        if ((jj & 3) < 2) {   // the k pairs whose channels sit in the lower half's registers
          const int i0 = 4 * (jj / 4) + 2 * (jj % 2);
          swap32(h2[i0], h2[i0 + 1], aop[jj], aop[jj + 2]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f16v d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = bias3[nt];
#pragma unroll
        for (int j = 0; j < H2 / 2; ++j) d = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[j], bw3[nt][j], d, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) tile[(32 * t + 8 * (i / 4) + 4 * half + (i % 4)) * TS + l32 + 32 * nt] = d[i];
      }
    }
    load_vtx(rnext, vnext);   // the next chunk's rows have arrived by now: what their vertices point to
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- the walk, 16 rows of the column at a time (independent LDS reads)
#pragma unroll
    for (int b0 = 0; b0 < 64; b0 += 16) {
      if (b0 < cnt) {   // wave-uniform
        float col[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) col[j] = tile[(b0 + j) * TS + c];
#pragma unroll
        for (int j = 0; j < 16; ++j) walk_row(cur, b0 + j, col[j]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile is rewritten by the next chunk
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    rin = rnext;
    vin = vnext;
  }
}

template <int CIN>
static int launch_pool_bins_mfma(const TlnBins& bn, int64_t rows, const float* const* w, const float* const* b, int min_points,
                                 unsigned long long* packed, float* d_out, int32_t* d_argrow, hipStream_t s) {
  const size_t lds = (size_t)(4 * (64 * 65)) * sizeof(float);
  const int64_t chunks = tln_cdiv(rows, 64);
  int64_t blocks = tln_cdiv(chunks, 4);
  if (blocks > 512) blocks = 512;   // two workgroups per CU: a wave takes several chunks on one set of weight registers
  auto kern = k_pool_bins_mfma<CIN>;
  static thread_local TlnLdsAttr attr;   // (one per template instantiation)
  TLN_HIP(tln_set_max_lds(attr, reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, bn, rows, min_points, w[0], b[0], w[1], b[1], w[2], b[2],
                     packed, d_out, d_argrow);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}


int tln_pool_bins_mfma_launch(int cin, const TlnBins& bn, int64_t rows, const float* const* w, const float* const* b,
                              int min_points, unsigned long long* packed, float* d_out, int32_t* d_argrow, hipStream_t s) {
  if (cin == 4) return launch_pool_bins_mfma<4>(bn, rows, w, b, min_points, packed, d_out, d_argrow, s);
  if (cin == 3) return launch_pool_bins_mfma<3>(bn, rows, w, b, min_points, packed, d_out, d_argrow, s);
  tln_set_error("matrix-core pool: cin %d", cin);
  return TLN_E_INVALID;
}
