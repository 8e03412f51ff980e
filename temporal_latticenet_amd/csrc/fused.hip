// K7 GroupNorm statistics, K9 GRU gates, K10 AFlow correlation, K8 slice, torch_scatter equivalents.
#include "common.h"
#include "gemm_args.h"
#include <math.h>

// =======================================================================================
// K7  GroupNorm over the lattice: statistics over (all vertices x channels of the group)
//     (Gn / GnRelu1x1 / GnReluConv / GnReluCoarsen / GnReluFinefy; reference use lm:75, lm:100)
// Two deterministic passes: per-row-block (sum, sumsq) in double, then a fixed-order combine.
// The result is per-CHANNEL scale/shift consumed by the gather-GEMM prologue.
// =======================================================================================
#define GN_ROWS_PER_BLOCK 32
#define GN_MAX_C 512

// grid (row blocks of 32, channel tiles of 64); block = 4 waves, wave w takes rows r0+w, r0+w+4, ... (8 rows):
// every load is independent, so a lattice level of a few thousand vertices still spreads over hundreds of blocks
// (up to TLN_FUSED_MAXJOBS tensors of one width per launch: blockIdx.z = tensor — the lock-stepped sequences of a stream)
#define TLN_FUSED_MAXJOBS 8
struct GnPartJobs {
  struct {
    const float* x;
    int64_t V;
    double2* partial;
  } j[TLN_FUSED_MAXJOBS];
};
__global__ void __launch_bounds__(256) k_gn_partial(const GnPartJobs jobs, int C) {
  const float* __restrict__ x = jobs.j[blockIdx.z].x;
  const int64_t V = jobs.j[blockIdx.z].V;
  double2* __restrict__ partial = jobs.j[blockIdx.z].partial;
  if ((int64_t)blockIdx.x * GN_ROWS_PER_BLOCK >= V) return;   // (the grid is sized for the tallest tensor)
  __shared__ double2 red[4][64];
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lx;
  const int64_t r0 = (int64_t)blockIdx.x * GN_ROWS_PER_BLOCK;
  float v[GN_ROWS_PER_BLOCK / 4];
#pragma unroll
  for (int i = 0; i < GN_ROWS_PER_BLOCK / 4; ++i) {
    const int64_t r = r0 + ly + 4 * i;
    v[i] = (r < V && c < C) ? x[r * C + c] : 0.0f;
  }
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int i = 0; i < GN_ROWS_PER_BLOCK / 4; ++i) {
    const double d = (double)v[i];
    s += d;
    q += d * d;
  }
  red[ly][lx] = make_double2(s, q);
  __syncthreads();
  if (ly == 0 && c < C) {
    double2 a = red[0][lx];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      a.x += red[w][lx].x;
      a.y += red[w][lx].y;
    }
    partial[(int64_t)blockIdx.x * C + c] = a;
  }
}

// one BLOCK per group: thread t adds the group's partial sums t, t+256, ... (fixed order), the 256 subtotals are
// combined by a fixed tree => deterministic.  (One wave per group, as this kernel used to be, walks 6k partials of a
// 30k-vertex level serially: 18 us; a block takes 3.)
struct GnFin {
  const double2* partial;
  int nblk;
  int64_t V;
  int C, groups;
  const float* gamma;
  const float* beta;
  float eps;
  float* scale;
  float* shift;
};
struct GnFinN {
  GnFin a[8];
};

__device__ __forceinline__ void gn_finalize_body(const GnFin& f, int g) {
  __shared__ double2 red[256];
  const double2* __restrict__ partial = f.partial;
  const int C = f.C;
  const int cpg = C / f.groups;
  const int64_t items = (int64_t)f.nblk * cpg;
  double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
  int64_t i = threadIdx.x;
  for (; i + 256 < items; i += 512) {          // two independent loads in flight per round
    const int64_t b0 = i / cpg, b1 = (i + 256) / cpg;
    const double2 p0 = partial[b0 * C + g * cpg + (int)(i - b0 * cpg)];
    const double2 p1 = partial[b1 * C + g * cpg + (int)(i + 256 - b1 * cpg)];
    s0 += p0.x;
    q0 += p0.y;
    s1 += p1.x;
    q1 += p1.y;
  }
  if (i < items) {
    const int64_t b0 = i / cpg;
    const double2 p0 = partial[b0 * C + g * cpg + (int)(i - b0 * cpg)];
    s0 += p0.x;
    q0 += p0.y;
  }
  red[threadIdx.x] = make_double2(s0 + s1, q0 + q1);
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[threadIdx.x].x += red[threadIdx.x + o].x;
      red[threadIdx.x].y += red[threadIdx.x + o].y;
    }
    __syncthreads();
  }
  const double s = red[0].x, q = red[0].y;
  const double cnt = (double)f.V * (double)cpg;
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)f.eps);
  for (int c = g * cpg + threadIdx.x; c < (g + 1) * cpg; c += 256) {
    const double gm = f.gamma ? (double)f.gamma[c] : 1.0;
    const double bt = f.beta ? (double)f.beta[c] : 0.0;
    f.scale[c] = (float)(gm * rstd);
    f.shift[c] = (float)(bt - mean * rstd * gm);
  }
}

__global__ void __launch_bounds__(256) k_gn_finalize(const GnFin f) { gn_finalize_body(f, blockIdx.x); }
// the same for the tensors of up to eight lock-stepped sequences in one launch (blockIdx.y = tensor)
__global__ void __launch_bounds__(256) k_gn_finalize_multi(const GnFinN ff) {
  const GnFin& f = ff.a[blockIdx.y];
  if ((int)blockIdx.x >= f.groups) return;
  gn_finalize_body(f, blockIdx.x);
}

extern "C" int64_t tln_groupnorm_ws_bytes(int64_t V, int C) {
  return tln_cdiv(V, GN_ROWS_PER_BLOCK) * (int64_t)C * (int64_t)sizeof(double2);
}

extern "C" int tln_groupnorm_stats(const float* d_x, int64_t V, int C, int groups, const float* d_gamma,
                                   const float* d_beta, float eps, float* d_scale, float* d_shift, void* d_ws,
                                   int64_t ws_bytes, void* stream_) {
  TLN_REQUIRE(d_x && d_scale && d_shift && d_ws, "null argument");
  TLN_REQUIRE(V > 0 && C > 0 && C <= GN_MAX_C && groups > 0 && C % groups == 0, "bad groupnorm shape V=%lld C=%d G=%d",
              (long long)V, C, groups);
  TLN_REQUIRE(ws_bytes >= tln_groupnorm_ws_bytes(V, C), "groupnorm workspace too small");
  hipStream_t s = (hipStream_t)stream_;
  const int nblk = (int)tln_cdiv(V, GN_ROWS_PER_BLOCK);
  GnPartJobs jobs;
  for (int i = 0; i < TLN_FUSED_MAXJOBS; ++i) {
    jobs.j[i].x = d_x;
    jobs.j[i].V = V;
    jobs.j[i].partial = (double2*)d_ws;
  }
  hipLaunchKernelGGL(k_gn_partial, dim3(nblk, (unsigned)tln_cdiv(C, 64), 1), dim3(256), 0, s, jobs, C);
  hipLaunchKernelGGL(k_gn_finalize, dim3((unsigned)groups), dim3(256), 0, s,
                     GnFin{(const double2*)d_ws, nblk, V, C, groups, d_gamma, d_beta, eps, d_scale, d_shift});
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

extern "C" int tln_groupnorm_partials(const float* d_x, int64_t V, int C, void* d_partials, void* stream_) {
  const tln_gn_partials_call c{d_x, V, d_partials};
  return tln_groupnorm_partials_multi(&c, 1, C, stream_);
}

// the partial sums of n tensors of one width (lock-stepped sequences) in one launch (blockIdx.z = tensor)
extern "C" int tln_groupnorm_partials_multi(const tln_gn_partials_call* c, int n, int C, void* stream_) {
  TLN_REQUIRE(c && n >= 1 && C > 0 && C <= GN_MAX_C, "bad groupnorm partials call");
  for (int i0 = 0; i0 < n; i0 += TLN_FUSED_MAXJOBS) {
    const int m = n - i0 < TLN_FUSED_MAXJOBS ? n - i0 : TLN_FUSED_MAXJOBS;
    GnPartJobs jobs;
    int64_t vmax = 0;
    for (int i = 0; i < TLN_FUSED_MAXJOBS; ++i) {
      const tln_gn_partials_call& a = c[i0 + (i < m ? i : 0)];
      TLN_REQUIRE(a.d_x && a.d_partials && a.V >= 0, "bad groupnorm partials call");
      jobs.j[i].x = a.d_x;
      jobs.j[i].V = a.V;
      jobs.j[i].partial = (double2*)a.d_partials;
      if (i < m && a.V > vmax) vmax = a.V;
    }
    if (vmax <= 0) continue;
    hipLaunchKernelGGL(k_gn_partial, dim3((unsigned)tln_cdiv(vmax, GN_ROWS_PER_BLOCK), (unsigned)tln_cdiv(C, 64), (unsigned)m),
                       dim3(256), 0, (hipStream_t)stream_, jobs, C);
    TLN_LAUNCH_CHECK();
  }
  return TLN_OK;
}

extern "C" int tln_groupnorm_from_partials(const void* d_partials, int64_t V, int C, int groups, const float* d_gamma,
                                           const float* d_beta, float eps, float* d_scale, float* d_shift,
                                           void* stream_) {
  TLN_REQUIRE(d_partials && d_scale && d_shift, "null argument");
  TLN_REQUIRE(V > 0 && C > 0 && groups > 0 && C % groups == 0, "bad groupnorm shape V=%lld C=%d G=%d", (long long)V, C,
              groups);
  const int nblk = (int)tln_cdiv(V, GN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(k_gn_finalize, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream_,
                     GnFin{(const double2*)d_partials, nblk, V, C, groups, d_gamma, d_beta, eps, d_scale, d_shift});
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// n <= 8 tensors (lock-stepped sequences) in one launch; the arguments as arrays
int tln_groupnorm_from_partials_multi(int n, const void* const* d_partials, const int64_t* V, const int* C, const int* groups,
                                      const float* const* d_gamma, const float* const* d_beta, const float* eps,
                                      float* const* d_scale, float* const* d_shift, void* stream_) {
  TLN_REQUIRE(n >= 1 && n <= 8, "bad number of tensors %d", n);
  GnFinN ff;
  int gmax = 0;
  for (int i = 0; i < 8; ++i) {
    const int k = i < n ? i : 0;
    TLN_REQUIRE(d_partials[k] && d_scale[k] && d_shift[k], "null argument");
    TLN_REQUIRE(V[k] > 0 && C[k] > 0 && groups[k] > 0 && C[k] % groups[k] == 0, "bad groupnorm shape V=%lld C=%d G=%d",
                (long long)V[k], C[k], groups[k]);
    ff.a[i] = GnFin{(const double2*)d_partials[k], (int)tln_cdiv(V[k], GN_ROWS_PER_BLOCK), V[k], C[k], groups[k], d_gamma[k],
                    d_beta[k], eps[k], d_scale[k], d_shift[k]};
    if (i < n && groups[k] > gmax) gmax = groups[k];
  }
  hipLaunchKernelGGL(k_gn_finalize_multi, dim3((unsigned)gmax, (unsigned)n), dim3(256), 0, (hipStream_t)stream_, ff);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

__global__ void __launch_bounds__(256) k_affine_act(const float* __restrict__ x, int64_t total, int C,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    int relu, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  float v = fmaf(x[i], scale[c], shift[c]);
  if (relu) v = fmaxf(v, 0.f);
  out[i] = v;
}

extern "C" int tln_affine_act(const float* d_x, int64_t V, int C, const float* d_scale, const float* d_shift, int relu,
                              float* d_out, void* stream_) {
  TLN_REQUIRE(d_x && d_scale && d_shift && d_out, "null argument");
  const int64_t total = V * C;
  if (total <= 0) return TLN_OK;
  hipLaunchKernelGGL(k_affine_act, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_, d_x, total,
                     C, d_scale, d_shift, relu, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// =======================================================================================
// K9  GRU gates (torch.nn.GRUCell semantics, reference lm:62):
//   r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1-z)*n + z*h
// gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh come from two gather-GEMM launches; rows of h
// beyond Vh are the zero padding of lm:59-60.
// =======================================================================================
__global__ void __launch_bounds__(256) k_gru_gates(const float* __restrict__ gi, const float* __restrict__ gh,
                                                   const float* __restrict__ h, int64_t V, int64_t Vh, int C,
                                                   float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * C) return;
  const int64_t v = i / C;
  const int c = (int)(i - v * C);
  const float* a = gi + v * 3 * C;
  const float* b = gh + v * 3 * C;
  const float hp = (v < Vh) ? h[v * C + c] : 0.0f;
  out[i] = tln_gru_cell_value(a[c] + b[c], a[C + c] + b[C + c], a[2 * C + c], b[2 * C + c], hp);   // common.h: one arithmetic
}

extern "C" int tln_gru_cell_opt(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const float* d_w_ih,
                                const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out, float* d_ws,
                                int64_t ws_floats, const tln_options* opt, void* stream_) {
  const tln_options& o = tln_opt(opt);
  TLN_REQUIRE(d_x && d_h && d_w_ih && d_w_hh && d_out && d_ws, "null argument");
  TLN_REQUIRE(V > 0 && Vh >= 0 && Vh <= V && C > 0, "bad GRU shape V=%lld Vh=%lld C=%d", (long long)V, (long long)Vh, C);
  TLN_REQUIRE(ws_floats >= V * 6 * (int64_t)C, "GRU workspace too small");
  float* gi = d_ws;
  float* gh = d_ws + V * 3 * (int64_t)C;
  tln_gemm_src sx{};
  sx.d_src = d_x;
  sx.src_rows = V;
  sx.ld = C;
  sx.cin = C;
  sx.taps = 1;
  tln_gemm_src sh = sx;
  sh.d_src = d_h;
  sh.src_rows = Vh;
  sh.pad_value = 0.0f;
  const tln_gemm_call ci{V, 3 * C, &sx, nullptr, d_w_ih, 1, d_b_ih, nullptr, 0, 0, gi, 3 * (int64_t)C, nullptr};
  // large lattices: the whole cell as ONE two-source product with the gates in its epilogue — neither gi nor gh in
  // memory, no gates kernel (gemm_v2.hip); rows and channels must be 16-byte aligned as for every gemm_v2 launch
  if (tln_gemm_v2_gru_ok(V, Vh, C, o) && d_b_ih && d_b_hh && ((uintptr_t)d_x % 16 == 0) && ((uintptr_t)d_h % 16 == 0) &&
      ((uintptr_t)d_w_ih % 16 == 0) && ((uintptr_t)d_w_hh % 16 == 0)) {
    int rc = tln_gemm_v2_launch_gru(d_x, d_h, Vh, V, C, d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_out, (hipStream_t)stream_);
    if (rc) return rc;
    TLN_LAUNCH_CHECK();
    return TLN_OK;
  }
  // the two products have the same shape: one launch (blockIdx.z = product) when they take the small-M kernel
  const tln_gemm_call ch{V, 3 * C, &sh, nullptr, d_w_hh, 1, d_b_hh, nullptr, 0, 0, gh, 3 * (int64_t)C, nullptr};
  const tln_gemm_call two[2] = {ci, ch};
  int rc = tln_gather_gemm_multi_opt(two, 2, &o, stream_);
  if (rc) return rc;
  const int64_t total = V * C;
  hipLaunchKernelGGL(k_gru_gates, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_, gi, gh, d_h,
                     V, Vh, C, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// the GRU cells of n lock-stepped sequences (same weights): ONE launch of the fused cell kernel (blockIdx.z = sequence)
// when every lattice is large enough for it, else one tln_gru_cell per sequence
extern "C" int tln_gru_cell(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const float* d_w_ih,
                            const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out, float* d_ws,
                            int64_t ws_floats, void* stream_) {
  return tln_gru_cell_opt(d_x, d_h, V, Vh, C, d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_out, d_ws, ws_floats, nullptr, stream_);
}

extern "C" int tln_gru_cell_multi_opt(const tln_gru_call* c, int n, int C, const float* d_w_ih, const float* d_w_hh,
                                      const float* d_b_ih, const float* d_b_hh, const tln_options* opt, void* stream_) {
  TLN_REQUIRE(c && n >= 1 && d_w_ih && d_w_hh && C > 0, "bad GRU batch");
  const tln_options& o = tln_opt(opt);
  bool fused = n >= 2 && n <= TLN_GEMM_MULTI_MAX && d_b_ih && d_b_hh && ((uintptr_t)d_w_ih % 16 == 0) &&
               ((uintptr_t)d_w_hh % 16 == 0);
  for (int i = 0; i < n && fused; ++i)
    fused = c[i].d_x && c[i].d_h && c[i].d_out && tln_gemm_v2_gru_ok(c[i].V, c[i].Vh, C, o) && ((uintptr_t)c[i].d_x % 16 == 0) &&
            ((uintptr_t)c[i].d_h % 16 == 0);
  if (fused) {
    const float* x[TLN_GEMM_MULTI_MAX];
    const float* h[TLN_GEMM_MULTI_MAX];
    float* out[TLN_GEMM_MULTI_MAX];
    int64_t V[TLN_GEMM_MULTI_MAX], Vh[TLN_GEMM_MULTI_MAX];
    for (int i = 0; i < n; ++i) {
      x[i] = c[i].d_x;
      h[i] = c[i].d_h;
      out[i] = c[i].d_out;
      V[i] = c[i].V;
      Vh[i] = c[i].Vh;
    }
    int rc = tln_gemm_v2_launch_gru_multi(n, x, h, Vh, V, C, d_w_ih, d_w_hh, d_b_ih, d_b_hh, out, (hipStream_t)stream_);
    if (rc) return rc;
    TLN_LAUNCH_CHECK();
    return TLN_OK;
  }
  for (int i = 0; i < n; ++i) {
    int rc = tln_gru_cell_opt(c[i].d_x, c[i].d_h, c[i].V, c[i].Vh, C, d_w_ih, d_w_hh, d_b_ih, d_b_hh, c[i].d_out, c[i].d_ws,
                              c[i].ws_floats, &o, stream_);
    if (rc) return rc;
  }
  return TLN_OK;
}

extern "C" int tln_gru_cell_multi(const tln_gru_call* c, int n, int C, const float* d_w_ih, const float* d_w_hh,
                                  const float* d_b_ih, const float* d_b_hh, void* stream_) {
  return tln_gru_cell_multi_opt(c, n, C, d_w_ih, d_w_hh, d_b_ih, d_b_hh, nullptr, stream_);
}

// =======================================================================================
// Element-wise steps of the alternative fusion modules (reference lm:17-185), so that every `rnn_modules` choice runs
// on this library's kernels end to end:
//   LSTMModule lm:36-38              h' = sig(o) * tanh(sig(i) * tanh(g))     (LSTMCell with a zero cell state)
//   TemporalMaxPoolModule lm:138-141 out = max(pad(h, -9999), x)
//   CrossframeGlobalAttentionModule lm:104-112   out = sig(a * s) * x, rows born in this frame pass unchanged
//   PointNetSeqModule lm:555-562     rows whose first half is all zero are filled with -9900 (early max-pool fusion)
// =======================================================================================
__device__ __forceinline__ float tln_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(256) k_lstm_gates(const float* __restrict__ gates, int64_t V, int C,
                                                    float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * C) return;
  const int64_t v = i / C;
  const int c = (int)(i - v * C);
  const float* g = gates + v * 4 * C;   // i | f | g | o (torch.nn.LSTMCell order); f multiplies the zero cell state
  const float cell = tln_sigmoid(g[c]) * tanhf(g[2 * C + c]);
  out[i] = tln_sigmoid(g[3 * C + c]) * tanhf(cell);
}

extern "C" int tln_lstm_gates(const float* d_gates, int64_t V, int C, float* d_out, void* stream_) {
  TLN_REQUIRE(d_gates && d_out && C > 0 && V >= 0, "bad LSTM gate arguments");
  if (V == 0) return TLN_OK;
  hipLaunchKernelGGL(k_lstm_gates, dim3((unsigned)tln_cdiv(V * C, 256)), dim3(256), 0, (hipStream_t)stream_, d_gates, V,
                     C, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

__global__ void __launch_bounds__(256) k_temporal_max(const float* __restrict__ x, const float* __restrict__ h,
                                                      int64_t V, int64_t Vh, int C, float pad,
                                                      float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * C) return;
  const float hv = (i / C < Vh) ? h[i] : pad;
  out[i] = fmaxf(hv, x[i]);
}

extern "C" int tln_temporal_max(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, float pad_value,
                                float* d_out, void* stream_) {
  TLN_REQUIRE(d_x && d_h && d_out && C > 0 && V >= 0 && Vh >= 0 && Vh <= V, "bad temporal max arguments");
  if (V == 0) return TLN_OK;
  hipLaunchKernelGGL(k_temporal_max, dim3((unsigned)tln_cdiv(V * C, 256)), dim3(256), 0, (hipStream_t)stream_, d_x, d_h,
                     V, Vh, C, pad_value, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

__global__ void __launch_bounds__(256) k_cga_gate(const float* __restrict__ a, const float* __restrict__ x, int64_t V,
                                                  int64_t Vh, int C, float scale, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * C) return;
  const float gte = (i / C < Vh) ? tln_sigmoid(a[i] * scale) : 1.0f;
  out[i] = gte * x[i];
}

extern "C" int tln_cga_gate(const float* d_a, const float* d_x, int64_t V, int64_t Vh, int C, float scale, float* d_out,
                            void* stream_) {
  TLN_REQUIRE(d_a && d_x && d_out && C > 0 && V >= 0 && Vh >= 0 && Vh <= V, "bad attention gate arguments");
  if (V == 0) return TLN_OK;
  hipLaunchKernelGGL(k_cga_gate, dim3((unsigned)tln_cdiv(V * C, 256)), dim3(256), 0, (hipStream_t)stream_, d_a, d_x, V,
                     Vh, C, scale, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// one wave per row: |x[:, :half]| summed; an all-zero first half marks an empty vertex
__global__ void __launch_bounds__(256) k_fill_empty_rows(const float* __restrict__ x, int64_t V, int C, int half,
                                                         float value, float* __restrict__ out) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (v >= V) return;
  const int lane = threadIdx.x & 63;
  float s = 0.0f;
  for (int c = lane; c < half; c += 64) s += fabsf(x[v * C + c]);
  s = tln_wave_sum(s);
  const bool empty = s == 0.0f;
  for (int c = lane; c < C; c += 64) out[v * C + c] = empty ? value : x[v * C + c];
}

extern "C" int tln_fill_empty_rows(const float* d_x, int64_t V, int C, int half, float value, float* d_out,
                                   void* stream_) {
  TLN_REQUIRE(d_x && d_out && C > 0 && half > 0 && half <= C && V >= 0, "bad fill arguments");
  if (V == 0) return TLN_OK;
  hipLaunchKernelGGL(k_fill_empty_rows, dim3((unsigned)tln_cdiv(V * 64, 256)), dim3(256), 0, (hipStream_t)stream_, d_x,
                     V, C, half, value, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// =======================================================================================
// K10 AFlow correlation (CustomKernelConvLatticeIm2RowModule.forward, reference lm:298-339).
// One wave per vertex: distances of the 9 (padded) hidden-state rows to the current centre feature,
// mask, row-normalise, w = (alpha - min(d, alpha)) * beta, mask, weighted sum of the rows, + bias.
// =======================================================================================
__device__ __forceinline__ float aflow_row_elem(const float* __restrict__ h, int64_t Vh, int C, int idx, int c,
                                                float pad) {
  if (idx < 0) return 0.0f;             // im2row zero row for a missing neighbour
  if (idx >= Vh) return pad;            // hidden state padded for vertices born in this frame (lm:215)
  return h[(int64_t)idx * C + c];
}

struct AflowJobs {
  struct {
    const float* x;
    const float* h;
    int64_t V, Vh;
    const int32_t* table;
    float* out;
    float* weights;
    int32_t* nbr_idx;
  } j[TLN_FUSED_MAXJOBS];
  int xcd;   // sequences dealt to the XCDs (common.h: tln_xcd_block)
};
__global__ void __launch_bounds__(256) k_aflow(const AflowJobs jobs, int C, float alpha, float beta, float pad,
                                               int use_center, const float* __restrict__ bias) {
  int bx, job;
  tln_xcd_block(jobs.xcd, bx, job);
  const auto& J = jobs.j[job];
  const float* __restrict__ x = J.x;
  const float* __restrict__ h = J.h;
  const int64_t V = J.V, Vh = J.Vh;
  const int32_t* __restrict__ table = J.table;
  float* __restrict__ out = J.out;
  float* __restrict__ weights = J.weights;
  int32_t* __restrict__ nbr_idx = J.nbr_idx;
  const int64_t v = ((int64_t)bx * blockDim.x + threadIdx.x) >> 6;
  if (v >= V) return;
  const int lane = threadIdx.x & 63;
  int idx[TLN_TAPS];
  float dist[TLN_TAPS];
#pragma unroll
  for (int t = 0; t < TLN_TAPS; ++t) idx[t] = table[v * TLN_TAPS + t];
#pragma unroll
  for (int t = 0; t < TLN_TAPS; ++t) {
    float acc = 0.0f;
    for (int c = lane; c < C; c += 64) {
      const float d = aflow_row_elem(h, Vh, C, idx[t], c, pad) - x[v * C + c];
      acc = fmaf(d, d, acc);
    }
    acc = tln_wave_sum(acc);
    dist[t] = sqrtf(acc) * ((idx[t] != -1) ? 1.0f : 0.0f);
  }
  if (!use_center) dist[TLN_TAPS - 1] = dist[TLN_TAPS - 1] * 0.0f;
  float rs = 0.0f;
#pragma unroll
  for (int t = 0; t < TLN_TAPS; ++t) rs += dist[t];
  float w[TLN_TAPS];
#pragma unroll
  for (int t = 0; t < TLN_TAPS; ++t) {
    const float dn = dist[t] * 1.0f / rs;  // (d * 1) / sum, as written at lm:321 (0/0 -> NaN is kept)
    float wt = (alpha - fminf(dn, alpha)) * beta;
    // torch.min propagates NaN, fminf does not
    if (dn != dn) wt = dn;
    wt = wt * ((idx[t] != -1) ? 1.0f : 0.0f);
    w[t] = wt;
  }
  if (!use_center) w[TLN_TAPS - 1] = w[TLN_TAPS - 1] * 0.0f;
  for (int c = lane; c < C; c += 64) {
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < TLN_TAPS; ++t) acc += aflow_row_elem(h, Vh, C, idx[t], c, pad) * w[t];
    if (bias) acc += bias[c];
    out[v * C + c] = acc;
  }
  if (lane < TLN_TAPS) {
    float wl = w[0];
    int il = idx[0];
#pragma unroll
    for (int t = 1; t < TLN_TAPS; ++t)
      if (lane == t) {
        wl = w[t];
        il = idx[t];
      }
    weights[v * TLN_TAPS + lane] = wl;
    nbr_idx[v * TLN_TAPS + lane] = il;
  }
}

extern "C" int tln_aflow(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const int32_t* d_table,
                         float alpha, float beta, float pad_value, int use_center, const float* d_bias, float* d_out,
                         float* d_weights, int32_t* d_nbr_idx, void* stream_) {
  const tln_aflow_call c{d_x, d_h, V, Vh, d_table, d_out, d_weights, d_nbr_idx};
  return tln_aflow_multi(&c, 1, C, alpha, beta, pad_value, use_center, d_bias, stream_);
}

// the AFlow correlations of n lock-stepped sequences (same alpha / beta / bias) in one launch (blockIdx.y = sequence)
extern "C" int tln_aflow_multi(const tln_aflow_call* c, int n, int C, float alpha, float beta, float pad_value,
                               int use_center, const float* d_bias, void* stream_) {
  TLN_REQUIRE(c && n >= 1 && C > 0, "bad AFlow batch");
  for (int i0 = 0; i0 < n; i0 += TLN_FUSED_MAXJOBS) {
    const int m = n - i0 < TLN_FUSED_MAXJOBS ? n - i0 : TLN_FUSED_MAXJOBS;
    AflowJobs jobs;
    jobs.xcd = tln_xcd_on();
    int64_t vmax = 0;
    for (int i = 0; i < TLN_FUSED_MAXJOBS; ++i) {
      const tln_aflow_call& a = c[i0 + (i < m ? i : 0)];
      TLN_REQUIRE(a.d_x && a.d_h && a.d_table && a.d_out && a.d_weights && a.d_nbr_idx, "null argument");
      TLN_REQUIRE(a.V > 0 && a.Vh >= 0, "bad AFlow shape");
      jobs.j[i].x = a.d_x;
      jobs.j[i].h = a.d_h;
      jobs.j[i].V = a.V;
      jobs.j[i].Vh = a.Vh;
      jobs.j[i].table = a.d_table;
      jobs.j[i].out = a.d_out;
      jobs.j[i].weights = a.d_weights;
      jobs.j[i].nbr_idx = a.d_nbr_idx;
      if (i < m && a.V > vmax) vmax = a.V;
    }
    hipLaunchKernelGGL(k_aflow, dim3((unsigned)tln_cdiv(vmax * 64, 256), (unsigned)m), dim3(256), 0, (hipStream_t)stream_,
                       jobs, C, alpha, beta, pad_value, use_center, d_bias);
    TLN_LAUNCH_CHECK();
  }
  return TLN_OK;
}

// =======================================================================================
// K8 slice (SliceFastCUDALatticeModule models.py:465, SliceLatticeModule)
// =======================================================================================
__global__ void __launch_bounds__(256) k_slice_gather(const float* __restrict__ lv, int64_t V, int cb,
                                                      const int32_t* __restrict__ indices,
                                                      const float* __restrict__ weights, int64_t n,
                                                      float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = 4 * (cb + 1);
  const int64_t p = gid / per;
  if (p >= n) return;
  const int k = (int)(gid - p * per);
  const int r = k / (cb + 1), c = k - r * (cb + 1);
  const int idx = indices[4 * p + r];
  const float w = weights[4 * p + r];
  float v = 0.0f;
  if (idx >= 0 && idx < V) v = (c < cb) ? w * lv[(int64_t)idx * cb + c] : w;
  out[gid] = v;
}

extern "C" int tln_slice_gather(const float* d_lv, int64_t V, int cb, const int32_t* d_indices, const float* d_weights,
                                int64_t n, float* d_out, void* stream_) {
  TLN_REQUIRE(d_lv && d_indices && d_weights && d_out && cb > 0, "null argument");
  if (n <= 0) return TLN_OK;
  const int64_t total = n * 4 * (cb + 1);
  hipLaunchKernelGGL(k_slice_gather, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_, d_lv, V,
                     cb, d_indices, d_weights, n, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

__global__ void __launch_bounds__(256) k_slice(const float* __restrict__ lv, int64_t V, int C,
                                               const int32_t* __restrict__ indices, const float* __restrict__ weights,
                                               const float* __restrict__ delta, const float* __restrict__ bias,
                                               int64_t n, float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t p = gid / C;
  if (p >= n) return;
  const int c = (int)(gid - p * C);
  float acc = 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = indices[4 * p + r];
    float w = weights[4 * p + r];
    if (delta) w += delta[4 * p + r];
    if (idx >= 0 && idx < V) acc = fmaf(w, lv[(int64_t)idx * C + c], acc);
  }
  out[gid] = bias ? acc + bias[c] : acc;
}

extern "C" int tln_slice(const float* d_lv, int64_t V, int C, const int32_t* d_indices, const float* d_weights,
                         const float* d_delta, const float* d_bias, int64_t n, float* d_out, void* stream_) {
  TLN_REQUIRE(d_lv && d_indices && d_weights && d_out && C > 0, "null argument");
  if (n <= 0) return TLN_OK;
  const int64_t total = n * C;
  hipLaunchKernelGGL(k_slice, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_, d_lv, V, C,
                     d_indices, d_weights, d_delta, d_bias, n, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// ---------------------------------------------------------------------------------------
// DeformSlice head in ONE kernel (SliceFastCUDALatticeModule, models.py:465): per point
//   g      = for r: [w_r * b[idx_r, :cb], w_r]                          (cb = 8: 36 features)
//   hdn    = relu(W_pre g)                                              Linear(36, 36, no bias)
//   dw     = W_dw hdn + b_dw                                            Linear(36, 4)
//   out    = sum_r (w_r + dw_r) * scores[idx_r] + bias
// instead of gather -> two per-point products over [N, 36] -> blend (three [N, 36] round trips through HBM and four
// launches).  One thread per point, the 36 x 36 + 4 x 36 weights are wave-uniform (scalar loads).
// ---------------------------------------------------------------------------------------
#define TLN_SLICE_MAX_C 64
struct SliceJobs {
  struct {
    const float* b;
    const float* scores;
    int64_t V, n;
    const int32_t* indices;
    const float* weights;
    float* out;
    float* logsm;
  } j[TLN_FUSED_MAXJOBS];
  int xcd;   // sequences dealt to the XCDs (common.h: tln_xcd_block)
};
template <int CB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_slice_deform(const SliceJobs jobs, int C,
                                                      const float* __restrict__ w_pre, const float* __restrict__ w_dw,
                                                      const float* __restrict__ b_dw, const float* __restrict__ bias) {
  int bx, job;
  tln_xcd_block(jobs.xcd, bx, job);
  const auto& J = jobs.j[job];
  const int64_t n = J.n, V = J.V;
  if ((int64_t)bx * 64 >= n) return;   // (the grid is sized for the largest cloud; uniform per block)
  const float* __restrict__ b = J.b;
  const float* __restrict__ scores = J.scores;
  const int32_t* __restrict__ indices = J.indices;
  const float* __restrict__ weights = J.weights;
  float* __restrict__ out = J.out;
  float* __restrict__ logsm = J.logsm;
  constexpr int G = 4 * (CB + 1);            // 36 gathered features
  constexpr int GP = (G + 3) / 4 * 4;        // LDS rows padded to 16-byte multiples
  constexpr int PPB = 64;                    // points per block: FOUR lanes per point, each owns G/4 hidden units
  constexpr int JQ = G / 4;
  static_assert(G % 4 == 0, "hidden units split over four lanes");
  __shared__ __attribute__((aligned(16))) float wp_s[G * GP];
  __shared__ float wd_s[4 * G];
  __shared__ float wr_s[PPB][4];
  __shared__ int idx_s[PPB][4];
  // the block's logits, for the fused log-softmax (models.py:467): [PPB][C | 1], sized by the launch — with the other
  // arrays 15 KB for 26 classes, so that eight workgroups share a CU and the 1875 of a 120k-point scan run in ONE round
  // (the kernel is a chain of dependent gathers and four barriers: latency, not arithmetic)
  extern __shared__ float lg_s[];
  const int LS = C | 1;
  __shared__ float mx_s[PPB], lse_s[PPB];
  for (int i = threadIdx.x; i < G * GP; i += blockDim.x) {
    const int j = i / GP, k = i - j * GP;
    wp_s[i] = k < G ? w_pre[j * G + k] : 0.0f;
  }
  for (int i = threadIdx.x; i < 4 * G; i += blockDim.x) wd_s[i] = w_dw[i];
  __syncthreads();
  const int pl = threadIdx.x >> 2, q = threadIdx.x & 3;     // point in block, quarter of the hidden layer
  const int64_t p0 = (int64_t)bx * PPB;
  const int64_t p = p0 + pl;
  if (p < n) {
    int idx[4];
    float w[4];
    float g[GP];
#pragma unroll
    for (int k = 0; k < GP; ++k) g[k] = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      idx[r] = indices[4 * p + r];
      w[r] = weights[4 * p + r];
      const bool ok = idx[r] >= 0 && idx[r] < V;
      if (ok) {
        const float4* br = reinterpret_cast<const float4*>(b + (int64_t)idx[r] * CB);
#pragma unroll
        for (int c4 = 0; c4 < CB / 4; ++c4) {
          const float4 v = br[c4];
          g[r * (CB + 1) + 4 * c4] = w[r] * v.x;
          g[r * (CB + 1) + 4 * c4 + 1] = w[r] * v.y;
          g[r * (CB + 1) + 4 * c4 + 2] = w[r] * v.z;
          g[r * (CB + 1) + 4 * c4 + 3] = w[r] * v.w;
        }
        g[r * (CB + 1) + CB] = w[r];
      }
    }
    // this lane's hidden units j = q*JQ .. q*JQ+JQ-1 (each: 36 FMAs in input order) and their 4 contributions
    float dw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int jj = 0; jj < JQ; ++jj) {
      const int j = q * JQ + jj;
      float h = 0.0f;
#pragma unroll
      for (int k4 = 0; k4 < GP; k4 += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(wp_s + j * GP + k4);
        h = fmaf(wv.x, g[k4], h);
        h = fmaf(wv.y, g[k4 + 1], h);
        h = fmaf(wv.z, g[k4 + 2], h);
        h = fmaf(wv.w, g[k4 + 3], h);
      }
      h = fmaxf(h, 0.0f);
#pragma unroll
      for (int r = 0; r < 4; ++r) dw[r] = fmaf(wd_s[r * G + j], h, dw[r]);
    }
    // the four quarters of a point sit in neighbouring lanes: fixed-order sum (q = 0,1,2,3) + bias
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d0 = __shfl(dw[r], (threadIdx.x & 60) + 0, 64), d1 = __shfl(dw[r], (threadIdx.x & 60) + 1, 64);
      const float d2 = __shfl(dw[r], (threadIdx.x & 60) + 2, 64), d3 = __shfl(dw[r], (threadIdx.x & 60) + 3, 64);
      dw[r] = b_dw[r] + ((d0 + d1) + (d2 + d3));
    }
    if (q == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wr_s[pl][r] = w[r] + dw[r];
        idx_s[pl][r] = (idx[r] >= 0 && idx[r] < V) ? idx[r] : -1;
      }
    }
  }
  __syncthreads();
  // blend: the block's points x C classes are written by consecutive threads (coalesced rows of `out`, and each
  // gathered scores row is read by C neighbouring threads)
  const int live = (n - p0) < (int64_t)PPB ? (int)(n - p0) : PPB;
  const int live_c = live * C;   // (32-bit index arithmetic: a 64-bit division per element costs more than the blend)
  for (int e = threadIdx.x; e < live_c; e += blockDim.x) {
    const int pt = e / C, c = e - pt * C;
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ix = idx_s[pt][r];
      if (ix >= 0) acc = fmaf(wr_s[pt][r], scores[(int64_t)ix * C + c], acc);
    }
    const float v = bias ? acc + bias[c] : acc;
    out[(p0 + pt) * C + c] = v;
    if (logsm) lg_s[pt * LS + c] = v;
  }
  if (logsm == nullptr) return;   // uniform
  // log-softmax over the classes of every point (what LNN_SEQ.forward returns beside the raw scores, models.py:466-468):
  // x - max - log(sum exp(x - max)); the statistics by FOUR lanes per point (classes q, q + 4, ..; combined in a fixed
  // order by two butterfly steps), then coalesced rows again
  __syncthreads();
  if (pl < live) {
    float mx = -INFINITY;
    for (int c = q; c < C; c += 4) mx = fmaxf(mx, lg_s[pl * LS + c]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.0f;
    for (int c = q; c < C; c += 4) sum += expf(lg_s[pl * LS + c] - mx);
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    if (q == 0) {
      mx_s[pl] = mx;
      lse_s[pl] = logf(sum);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < live_c; e += blockDim.x) {
    const int pt = e / C, c = e - pt * C;
    logsm[(p0 + pt) * C + c] = (lg_s[pt * LS + c] - mx_s[pt]) - lse_s[pt];
  }
}

extern "C" int tln_slice_deform_ls(const float* d_b, int cb, const float* d_scores, int64_t V, int C,
                                   const int32_t* d_indices, const float* d_weights, const float* d_w_pre,
                                   const float* d_w_dw, const float* d_b_dw, const float* d_bias, int64_t n, float* d_out,
                                   float* d_logsm, void* stream_) {
  const tln_slice_call c{d_b, d_scores, V, d_indices, d_weights, n, d_out, d_logsm};
  return tln_slice_deform_multi(&c, 1, cb, C, d_w_pre, d_w_dw, d_b_dw, d_bias, stream_);
}

// the DeformSlice heads of n lock-stepped sequences (same head weights) in one launch (blockIdx.y = sequence); either
// every call asks for the log-softmax or none does
extern "C" int tln_slice_deform_multi(const tln_slice_call* c, int n, int cb, int C, const float* d_w_pre,
                                      const float* d_w_dw, const float* d_b_dw, const float* d_bias, void* stream_) {
  TLN_REQUIRE(c && n >= 1 && d_w_pre && d_w_dw && d_b_dw && C > 0, "null argument");
  TLN_REQUIRE(cb == 8, "the deform head is built for the 8-channel bottleneck (got %d)", cb);
  for (int i0 = 0; i0 < n; i0 += TLN_FUSED_MAXJOBS) {
    const int m = n - i0 < TLN_FUSED_MAXJOBS ? n - i0 : TLN_FUSED_MAXJOBS;
    SliceJobs jobs;
    jobs.xcd = tln_xcd_on();
    int64_t nmax = 0;
    const bool ls = c[i0].d_logsm != nullptr;
    for (int i = 0; i < TLN_FUSED_MAXJOBS; ++i) {
      const tln_slice_call& a = c[i0 + (i < m ? i : 0)];
      TLN_REQUIRE(a.d_b && a.d_scores && a.d_indices && a.d_weights && a.d_out, "null argument");
      TLN_REQUIRE((a.d_logsm != nullptr) == ls, "log-softmax asked for by some calls of a batch only");
      jobs.j[i].b = a.d_b;
      jobs.j[i].scores = a.d_scores;
      jobs.j[i].V = a.V;
      jobs.j[i].n = a.n;
      jobs.j[i].indices = a.d_indices;
      jobs.j[i].weights = a.d_weights;
      jobs.j[i].out = a.d_out;
      jobs.j[i].logsm = a.d_logsm;
      if (i < m && a.n > nmax) nmax = a.n;
    }
    TLN_REQUIRE(!ls || C <= TLN_SLICE_MAX_C, "fused log-softmax: at most %d classes (got %d)", TLN_SLICE_MAX_C, C);
    if (nmax <= 0) continue;
    const size_t lds = ls ? (size_t)64 * (C | 1) * sizeof(float) : 0;
    hipLaunchKernelGGL(k_slice_deform<8>, dim3((unsigned)tln_cdiv(nmax, 64), (unsigned)m), dim3(256), lds,
                       (hipStream_t)stream_, jobs, C, d_w_pre, d_w_dw, d_b_dw, d_bias);
    TLN_LAUNCH_CHECK();
  }
  return TLN_OK;
}

extern "C" int tln_slice_deform(const float* d_b, int cb, const float* d_scores, int64_t V, int C,
                                const int32_t* d_indices, const float* d_weights, const float* d_w_pre,
                                const float* d_w_dw, const float* d_b_dw, const float* d_bias, int64_t n, float* d_out,
                                void* stream_) {
  return tln_slice_deform_ls(d_b, cb, d_scores, V, C, d_indices, d_weights, d_w_pre, d_w_dw, d_b_dw, d_bias, n, d_out,
                             nullptr, stream_);
}

// =======================================================================================
// torch_scatter 2.0.4 equivalents (reference lm:485-520, models.py:454); dim=0 only
// =======================================================================================
__global__ void __launch_bounds__(256) k_scatter_max_pack(const float* __restrict__ src,
                                                          const int64_t* __restrict__ index, int64_t rows, int C,
                                                          int64_t out_rows, unsigned long long* __restrict__ packed) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = gid / C;
  if (r >= rows) return;
  const int c = (int)(gid - r * C);
  const int64_t o = index[r];
  if (o < 0 || o >= out_rows) return;
  const unsigned long long p = ((unsigned long long)tln_f2ord(src[gid]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)r);
  atomicMax(&packed[o * C + c], p);
}

__global__ void __launch_bounds__(256) k_scatter_max_unpack(const unsigned long long* __restrict__ packed, int64_t total,
                                                            int64_t rows, float* __restrict__ out,
                                                            int64_t* __restrict__ argmax) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const unsigned long long p = packed[gid];
  if (p == 0ull) {
    out[gid] = 0.0f;
    if (argmax) argmax[gid] = rows;
  } else {
    out[gid] = tln_ord2f((uint32_t)(p >> 32));
    if (argmax) argmax[gid] = (int64_t)(0xFFFFFFFFu - (uint32_t)(p & 0xFFFFFFFFull));
  }
}

extern "C" int tln_scatter_max(const float* d_src, const int64_t* d_index, int64_t rows, int C, int64_t out_rows,
                               float* d_out, int64_t* d_argmax, void* d_ws, int64_t ws_bytes, void* stream_) {
  TLN_REQUIRE(d_src && d_index && d_out && d_ws && C > 0, "null argument");
  TLN_REQUIRE(ws_bytes >= out_rows * C * 8, "scatter_max workspace too small");
  TLN_REQUIRE(rows < (1ll << 32), "too many rows");
  hipStream_t s = (hipStream_t)stream_;
  const int64_t total = out_rows * C;
  if (total <= 0) return TLN_OK;
  TLN_HIP(hipMemsetAsync(d_ws, 0, (size_t)total * 8, s));
  if (rows > 0)
    hipLaunchKernelGGL(k_scatter_max_pack, dim3((unsigned)tln_cdiv(rows * C, 256)), dim3(256), 0, s, d_src, d_index,
                       rows, C, out_rows, (unsigned long long*)d_ws);
  hipLaunchKernelGGL(k_scatter_max_unpack, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, s,
                     (const unsigned long long*)d_ws, total, rows, d_out, d_argmax);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

__global__ void __launch_bounds__(256) k_scatter_add(const float* __restrict__ src, const int64_t* __restrict__ index,
                                                     int64_t rows, int C, int64_t out_rows, float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = gid / C;
  if (r >= rows) return;
  const int c = (int)(gid - r * C);
  const int64_t o = index[r];
  if (o < 0 || o >= out_rows) return;
  atomicAdd(&out[o * C + c], src[gid]);
}

// d_out must be zero-initialised (or hold the `out=` tensor of torch_scatter) by the caller
extern "C" int tln_scatter_add(const float* d_src, const int64_t* d_index, int64_t rows, int C, int64_t out_rows,
                               float* d_out, void* stream_) {
  TLN_REQUIRE(d_src && d_index && d_out && C > 0, "null argument");
  if (rows <= 0) return TLN_OK;
  hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)tln_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream_, d_src,
                     d_index, rows, C, out_rows, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// =======================================================================================
// composite: GroupNorm statistics (from the producer's partial sums when they exist, else two passes over x)
// followed by the gather-GEMM that applies them in its operand staging.  One call from the host instead of two or
// three: on a lattice of a few thousand vertices the host-side launch path is the bottleneck, not the kernels.
// =======================================================================================
extern "C" int tln_gn_gather_gemm_opt(const tln_gn_desc* gn, int64_t M, int N, const tln_gemm_src* s0,
                                      const tln_gemm_src* s1, const float* d_w, int w_is_nk, const float* d_bias,
                                      const float* d_residual, int64_t ld_res, int relu, float* d_out, int64_t ld_out,
                                      void* d_stats, const tln_options* opt, void* stream_) {
  TLN_REQUIRE(gn && s0, "null argument");
  const void* partials = gn->d_partials;
  if (!partials) {  // the tensor did not come out of a gather-GEMM: one pass for the partial sums
    TLN_REQUIRE(gn->d_x && gn->d_ws && gn->ws_bytes >= tln_groupnorm_ws_bytes(gn->V, gn->C), "GroupNorm workspace");
    int rc = tln_groupnorm_partials(gn->d_x, gn->V, gn->C, gn->d_ws, stream_);
    if (rc) return rc;
    partials = gn->d_ws;
  }
  tln_gemm_src a = *s0;
  TLN_REQUIRE(a.cin == gn->C, "GroupNorm width %d does not match the GEMM source width %d", gn->C, a.cin);
  a.d_scale = gn->d_scale_shift;            // scratch for the large-V fallback (may be NULL when not needed)
  a.d_shift = gn->d_scale_shift ? gn->d_scale_shift + gn->C : nullptr;
  a.relu = gn->relu;
  a.d_gn_partials = partials;
  a.d_gn_gamma = gn->d_gamma;
  a.d_gn_beta = gn->d_beta;
  a.gn_rows = gn->V;
  a.gn_groups = gn->groups;
  a.gn_eps = gn->eps;
  const tln_gemm_call call{M, N, &a, s1, d_w, w_is_nk, d_bias, d_residual, ld_res, relu, d_out, ld_out, d_stats};
  return tln_gather_gemm_opt(&call, opt, stream_);
}

extern "C" int tln_gn_gather_gemm(const tln_gn_desc* gn, int64_t M, int N, const tln_gemm_src* s0,
                                  const tln_gemm_src* s1, const float* d_w, int w_is_nk, const float* d_bias,
                                  const float* d_residual, int64_t ld_res, int relu, float* d_out, int64_t ld_out,
                                  void* d_stats, void* stream_) {
  return tln_gn_gather_gemm_opt(gn, M, N, s0, s1, d_w, w_is_nk, d_bias, d_residual, ld_res, relu, d_out, ld_out, d_stats,
                                nullptr, stream_);
}
