// Backward of the gather-GEMM (training path; the reference trains through the un-vendored
// lattice_net autograd functions, train_ln.py:212-233):
//
//   forward   out[m, n] = sum_k sum_c  src[table[m, k], c] * W[k * cin + c, n]          (csrc/gemm*.hip)
//   dA        for a level's own neighbour table the taps come in pairs (tap k of row v is tap k^1 of its neighbour, the
//             centre is its own pair): dsrc[j, c] = sum_k sum_n dout[table[j, k^1], n] * W[k * cin + c, n] — again a
//             gather-GEMM over the same table, with the weight blocks of paired taps swapped and transposed.  That
//             product runs on the forward kernels (autograd.py builds the [9 N, cin] weight); nothing is scattered,
//             so dA is deterministic.  1x1 products: dsrc = dout @ W^T, the forward kernel with the other weight layout.
//   dW        dW[k * cin + c, n] = sum_m src[table[m, k], c] * dout[m, n]: THIS file.  A "TN" product whose long
//             dimension is M: one wave per (tap, 32 input channels, 32*TN output channels, slice of M) accumulates
//             its 32 x 32*TN tile with v_mfma_f32_32x32x2_f32 (A operand = the gathered source rows transposed: lane
//             = channel, k = two consecutive rows m; B operand = the dout rows), straight from global memory
//             (128-byte coalesced row pieces; every operand element feeds one MFMA per column tile).  The M slices
//             write partial tiles to a workspace, k_dw_reduce adds them in slice order: no atomics, the same bits on
//             every run.
#include "common.h"

typedef float f32x16b __attribute__((ext_vector_type(16)));

struct DwArgs {
  const float* src;
  const int32_t* table;   // [M, taps] or NULL (row m itself)
  const float* dout;
  float* ws;              // [splits][taps * cin][N]
  int64_t src_rows, M, rows_per_split;
  int cin, N, taps, splits;
};

template <int TN>
__global__ void __launch_bounds__(256) k_gather_gemm_dw(const DwArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int cblocks = (a.cin + 31) >> 5;
  const int tap = blockIdx.x / cblocks, c0 = (blockIdx.x - tap * cblocks) << 5;
  const int n0 = (blockIdx.y * 4 + wv) * 32 * TN;
  if (n0 >= a.N) return;
  const int64_t m_begin = (int64_t)blockIdx.z * a.rows_per_split;
  int64_t m_end = m_begin + a.rows_per_split;
  if (m_end > a.M) m_end = a.M;
  f32x16b acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  const float* __restrict__ src = a.src;
  const float* __restrict__ dout = a.dout;
  const int32_t* __restrict__ table = a.table;
  constexpr int U = 4;   // m pairs in flight per step: every load of a step is issued before its first MFMA
  for (int64_t m0 = m_begin; m0 < m_end; m0 += 2 * U) {
    float av[U], bv[U][TN];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = m0 + 2 * u + half;
      const bool live = m < m_end;
      int64_t idx = -1;
      if (live) idx = table ? (int64_t)table[m * a.taps + tap] : m;
      const bool has = idx >= 0 && idx < a.src_rows;
      av[u] = (has && c0 + l31 < a.cin) ? src[idx * a.cin + c0 + l31] : 0.0f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + 32 * j + l31;
        bv[u][j] = (live && n < a.N) ? dout[m * a.N + n] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][j], acc[j], 0, 0, 0);
  }
  // C/D layout: col = lane & 31 (n), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel)
  float* __restrict__ out = a.ws + (size_t)blockIdx.z * ((size_t)a.taps * a.cin * a.N);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + 32 * j + l31;
    if (n >= a.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (c < a.cin) out[((size_t)tap * a.cin + c) * a.N + n] = acc[j][r];
    }
  }
}

// dW[i] = ws[0][i] + ws[1][i] + ... in slice order (fixed: deterministic)
__global__ void __launch_bounds__(256) k_dw_reduce(const float* __restrict__ ws, int splits, int64_t elems,
                                                   float* __restrict__ dw) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float s = 0.0f;
  for (int k = 0; k < splits; ++k) s += ws[(size_t)k * elems + i];
  dw[i] = s;
}

static void dw_geometry(int64_t M, int cin, int taps, int N, int* tn, int* splits, int64_t* rows_per_split) {
  *tn = (N % 64 == 0) ? 2 : 1;
  const int64_t waves = (int64_t)taps * tln_cdiv(cin, 32) * tln_cdiv(N, 32 * *tn);
  int64_t s = tln_cdiv(4096, waves);           // ~4 waves per SIMD over the chip
  const int64_t max_s = tln_cdiv(M, 256);      // a slice has at least 256 rows
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  int64_t rps = tln_cdiv(M, s);
  rps = (rps + 7) / 8 * 8;
  *rows_per_split = rps;
  *splits = (int)tln_cdiv(M, rps);
}

extern "C" int64_t tln_gather_gemm_dw_ws_floats(int64_t M, int cin, int taps, int N) {
  if (M <= 0 || cin <= 0 || taps <= 0 || N <= 0) return 0;
  int tn, splits;
  int64_t rps;
  dw_geometry(M, cin, taps, N, &tn, &splits, &rps);
  return (int64_t)splits * taps * cin * N;
}

extern "C" int tln_gather_gemm_dw(const float* d_src, int64_t src_rows, int cin, const int32_t* d_table, int taps,
                                  const float* d_dout, int64_t M, int N, float* d_dw, float* d_ws, int64_t ws_floats,
                                  void* stream_) {
  TLN_REQUIRE(d_src && d_dout && d_dw && d_ws, "null argument");
  TLN_REQUIRE(M > 0 && N > 0 && cin > 0 && (taps == 1 || taps == TLN_TAPS), "bad dW shape");
  TLN_REQUIRE(taps == 1 || d_table, "a 9-tap product needs its table");
  int tn, splits;
  int64_t rps;
  dw_geometry(M, cin, taps, N, &tn, &splits, &rps);
  const int64_t elems = (int64_t)taps * cin * N;
  TLN_REQUIRE(ws_floats >= (int64_t)splits * elems, "dW workspace too small");
  hipStream_t s = (hipStream_t)stream_;
  DwArgs a{d_src, taps == 1 ? nullptr : d_table, d_dout, d_ws, src_rows, M, rps, cin, N, taps, splits};
  dim3 grid((unsigned)(taps * tln_cdiv(cin, 32)), (unsigned)tln_cdiv(N, 128 * tn), (unsigned)splits);
  if (tn == 2) hipLaunchKernelGGL(k_gather_gemm_dw<2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_gather_gemm_dw<1>, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)tln_cdiv(elems, 256)), dim3(256), 0, s, d_ws, splits, elems, d_dw);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// ======================================================================================================================
// Backward of the slice blends (SliceFastCUDALatticeModule / SliceLatticeModule, models.py:465): what autograd through
// `lv[idx] * w` does with index_add_ (float atomics in arrival order) done as a SEGMENT sum over the lattice's
// vertex-sorted row list (tln_build_csr: stable, so the order of a vertex's rows is the row order): one wave per vertex
// walks its rows in that order, lanes = channels.  Deterministic, no [N, 4, C] temporaries.
//   tln_slice_blend_bwd_lv      d_lv[v, c] = sum over rows of v of (w_row + delta_row) * dout[row >> 2, c]
//                                (per_row != 0: the value row is dvals[row] instead of dvals[row >> 2] — the backward of
//                                 tln_slice_gather, whose [N, 4 * (cb + 1)] output is [4N, cb + 1] row by row)
//   tln_slice_blend_bwd_w       d_w[row] = dot(lv[idx_row, :], dout[row >> 2, :])   (0 for rows without a vertex)
// ======================================================================================================================
struct tln_lattice;
const int32_t* tln_lat_order(const tln_lattice* l);
const int32_t* tln_lat_seg_start(const tln_lattice* l);
int64_t tln_lat_csr_rows(const tln_lattice* l);

__global__ void __launch_bounds__(256) k_slice_bwd_lv(const float* __restrict__ dvals, int64_t ld, int C, int per_row,
                                                      const float* __restrict__ weights, const float* __restrict__ delta,
                                                      const int32_t* __restrict__ order,
                                                      const int32_t* __restrict__ seg_start, int64_t nv,
                                                      float* __restrict__ out) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (v >= nv) return;
  const int lane = threadIdx.x & 63;
  const int b = seg_start[v], e = seg_start[v + 1];
  for (int c0 = 0; c0 < C; c0 += 256) {      // four channels per lane and pass
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int j = b; j < e; ++j) {
      const int64_t row = order[j];
      float w = weights[row];
      if (delta) w += delta[row];
      const float* x = dvals + (per_row ? row : (row >> 2)) * ld;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + 64 * k + lane;
        if (c < C) acc[k] = fmaf(w, x[c], acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + 64 * k + lane;
      if (c < C) out[v * C + c] = acc[k];
    }
  }
}

__global__ void __launch_bounds__(256) k_slice_bwd_w(const float* __restrict__ lv, int64_t V, int C,
                                                     const int32_t* __restrict__ indices, const float* __restrict__ dout,
                                                     int64_t rows, float* __restrict__ dw) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int idx = indices[row];
  float acc = 0.0f;
  if (idx >= 0 && idx < V) {
    const float* a = lv + (int64_t)idx * C;
    const float* g = dout + (row >> 2) * C;
    for (int c = lane; c < C; c += 64) acc = fmaf(a[c], g[c], acc);
  }
  acc = tln_wave_sum(acc);   // fixed butterfly: the same bits on every run
  if (lane == 0) dw[row] = acc;
}

extern "C" int tln_slice_blend_bwd_lv(tln_lattice_t* l, const float* d_dvals, int64_t ld, int C, int per_row,
                                      const float* d_weights, const float* d_delta, int64_t rows, float* d_out,
                                      void* stream_) {
  TLN_REQUIRE(l && d_dvals && d_weights && d_out && C > 0 && ld >= C, "null argument");
  TLN_REQUIRE(tln_lat_csr_rows(l) == rows, "the slice backward needs the CSR of tln_build_csr over the same %lld rows",
              (long long)rows);
  const int64_t nv = tln_lattice_nr_vertices(l);
  if (nv <= 0) return TLN_OK;
  hipLaunchKernelGGL(k_slice_bwd_lv, dim3((unsigned)tln_cdiv(nv * 64, 256)), dim3(256), 0, (hipStream_t)stream_, d_dvals, ld,
                     C, per_row, d_weights, d_delta, tln_lat_order(l), tln_lat_seg_start(l), nv, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

extern "C" int tln_slice_blend_bwd_w(const float* d_lv, int64_t V, int C, const int32_t* d_indices, const float* d_dout,
                                     int64_t rows, float* d_dw, void* stream_) {
  TLN_REQUIRE(d_lv && d_indices && d_dout && d_dw && C > 0, "null argument");
  if (rows <= 0) return TLN_OK;
  hipLaunchKernelGGL(k_slice_bwd_w, dim3((unsigned)tln_cdiv(rows * 64, 256)), dim3(256), 0, (hipStream_t)stream_, d_lv, V, C,
                     d_indices, d_dout, rows, d_dw);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}
