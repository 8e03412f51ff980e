// K3+K4: gather-GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
// One kernel serves ConvLatticeModule (9 taps through the neighbour table, reference
// lattice_modules.py:301/440/573), the coarsen / finefy cross-level convolutions (models.py:353, 398),
// every 1x1 linear (Conv1x1, GnRelu1x1, hidden_linear lm:47, AFlow linear lm:196) and two-source
// products (cat([a,b]) @ W, lm:223-226).  The [V, 9*C] im2row matrix of the reference is never written:
// the A operand is gathered row by row through the table straight into LDS, with the preceding
// GroupNorm-apply + ReLU folded into the staging (y = relu(x*scale[c] + shift[c]); a missing neighbour
// stays an exact zero row, as in the reference where im2row pads AFTER the activation).
//
// Tiling: 256 threads = 2x2 waves, each wave owns TMxTN tiles of 32x32 (block = 64TM x 64TN), K stepped
// in BK chunks that never straddle a tap.  LDS tiles are k-major (As[k][m], Bs[k][n]) so that the MFMA
// operand reads (lane -> m|n = lane&31, k = lane>>5) are conflict-free ds_read_b32; the next chunk's global
// loads are issued before the MFMAs of the current one (register prefetch, one barrier pair per chunk).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SrcDev {
  const float* src;
  const int32_t* table;
  const float* scale;
  const float* shift;
  int64_t src_rows, ld;
  int cin, taps, relu;
  float pad;
};

struct GemmArgs {
  int64_t M;
  int N, K0;
  SrcDev s[2];
  int nsrc;
  const float* W;
  int64_t ldw;
  const float* bias;
  const float* res;
  int64_t ld_res;
  int relu;
  float* out;
  int64_t ld_out;
};

template <int TM, int TN, int BK, bool W_NK, bool VEC>
__global__ void __launch_bounds__(256) k_gather_gemm(const GemmArgs g) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDA = BM + 1;
  constexpr int LDB = W_NK ? BN + 1 : BN;
  constexpr int KQ = BK / 4;                 // float4 per row chunk
  constexpr int A_ROWS_PASS = 256 / KQ;      // rows staged per pass
  constexpr int A_PASSES = BM / A_ROWS_PASS;
  constexpr int B_F4_ROW = BN / 4;
  constexpr int B_ROWS_PASS = 256 / B_F4_ROW;  // [K,N] layout: k rows per pass
  constexpr int B_PASSES_KN = BK / B_ROWS_PASS;
  constexpr int B_PASSES_NK = BN / A_ROWS_PASS;  // [N,K] layout: n rows per pass (same shape as A)
  constexpr int B_PASSES = W_NK ? B_PASSES_NK : B_PASSES_KN;
  static_assert(A_PASSES >= 1 && B_PASSES >= 1, "tile too small for 256 threads");

  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB + 4];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int cpt0 = (g.s[0].cin + BK - 1) / BK;
  const int nch0 = g.s[0].taps * cpt0;
  const int cpt1 = g.nsrc > 1 ? (g.s[1].cin + BK - 1) / BK : 1;
  const int nch1 = g.nsrc > 1 ? g.s[1].taps * cpt1 : 0;
  const int nchunks = nch0 + nch1;

  const int a_kq = tid % KQ;
  const int a_row0 = tid / KQ;

  float4 a_reg[A_PASSES];
  float4 b_reg[B_PASSES];

  auto prefetch = [&](int t) {
    const int si = (t < nch0) ? 0 : 1;
    const SrcDev& s = g.s[si];
    const int tt = si ? t - nch0 : t;
    const int cpt = si ? cpt1 : cpt0;
    const int tap = tt / cpt;
    const int c0 = (tt - tap * cpt) * BK;
    const int kvalid = (s.cin - c0) < BK ? (s.cin - c0) : BK;
    const int kbase = (si ? g.K0 : 0) + tap * s.cin + c0;
    // ---- A: gathered rows
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int64_t m = m0 + p * A_ROWS_PASS + a_row0;
      const int c = c0 + 4 * a_kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < g.M && 4 * a_kq < kvalid) {
        const int64_t srow = s.table ? (int64_t)s.table[m * s.taps + tap] : m;
        if (srow >= 0) {
          if (srow >= s.src_rows) {
            v = make_float4(s.pad, s.pad, s.pad, s.pad);
          } else {
            const float* ptr = s.src + srow * s.ld + c;
            if (VEC) {
              v = *reinterpret_cast<const float4*>(ptr);
            } else {
              v.x = ptr[0];
              if (c + 1 < s.cin) v.y = ptr[1];
              if (c + 2 < s.cin) v.z = ptr[2];
              if (c + 3 < s.cin) v.w = ptr[3];
            }
            if (s.scale) {
              float sc[4], sh[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const bool ok = VEC || (c + j < s.cin);
                sc[j] = ok ? s.scale[c + j] : 0.f;
                sh[j] = ok ? s.shift[c + j] : 0.f;
              }
              v.x = fmaf(v.x, sc[0], sh[0]);
              v.y = fmaf(v.y, sc[1], sh[1]);
              v.z = fmaf(v.z, sc[2], sh[2]);
              v.w = fmaf(v.w, sc[3], sh[3]);
            }
            if (s.relu) {
              v.x = fmaxf(v.x, 0.f);
              v.y = fmaxf(v.y, 0.f);
              v.z = fmaxf(v.z, 0.f);
              v.w = fmaxf(v.w, 0.f);
            }
            if (!VEC) {  // columns past cin must stay zero
              if (c + 1 >= s.cin) v.y = 0.f;
              if (c + 2 >= s.cin) v.z = 0.f;
              if (c + 3 >= s.cin) v.w = 0.f;
            }
          }
        }
      }
      a_reg[p] = v;
    }
    // ---- B: weights
    if (!W_NK) {
      const int nq = tid % B_F4_ROW, kr0 = tid / B_F4_ROW;
#pragma unroll
      for (int p = 0; p < B_PASSES; ++p) {
        const int k = p * B_ROWS_PASS + kr0;
        const int n = n0 + 4 * nq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kvalid && n < g.N) {
          const float* ptr = g.W + (int64_t)(kbase + k) * g.ldw + n;
          if (VEC) {
            v = *reinterpret_cast<const float4*>(ptr);
          } else {
            v.x = ptr[0];
            if (n + 1 < g.N) v.y = ptr[1];
            if (n + 2 < g.N) v.z = ptr[2];
            if (n + 3 < g.N) v.w = ptr[3];
          }
        }
        b_reg[p] = v;
      }
    } else {
#pragma unroll
      for (int p = 0; p < B_PASSES; ++p) {
        const int n = n0 + p * A_ROWS_PASS + a_row0;
        const int kk = 4 * a_kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < g.N && kk < kvalid) {
          const float* ptr = g.W + (int64_t)n * g.ldw + kbase + kk;
          if (VEC) {
            v = *reinterpret_cast<const float4*>(ptr);
          } else {
            v.x = ptr[0];
            if (kk + 1 < kvalid) v.y = ptr[1];
            if (kk + 2 < kvalid) v.z = ptr[2];
            if (kk + 3 < kvalid) v.w = ptr[3];
          }
        }
        b_reg[p] = v;
      }
    }
  };

  auto stage = [&]() {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int row = p * A_ROWS_PASS + a_row0;
      float* d = As + (4 * a_kq) * LDA + row;
      d[0] = a_reg[p].x;
      d[LDA] = a_reg[p].y;
      d[2 * LDA] = a_reg[p].z;
      d[3 * LDA] = a_reg[p].w;
    }
    if (!W_NK) {
      const int nq = tid % B_F4_ROW, kr0 = tid / B_F4_ROW;
#pragma unroll
      for (int p = 0; p < B_PASSES; ++p) {
        const int k = p * B_ROWS_PASS + kr0;
        *reinterpret_cast<float4*>(Bs + k * LDB + 4 * nq) = b_reg[p];
      }
    } else {
#pragma unroll
      for (int p = 0; p < B_PASSES; ++p) {
        const int nrow = p * A_ROWS_PASS + a_row0;
        float* d = Bs + (4 * a_kq) * LDB + nrow;
        d[0] = b_reg[p].x;
        d[LDB] = b_reg[p].y;
        d[2 * LDB] = b_reg[p].z;
        d[3 * LDB] = b_reg[p].w;
      }
    }
  };

  prefetch(0);
  for (int t = 0; t < nchunks; ++t) {
    stage();
    __syncthreads();
    if (t + 1 < nchunks) prefetch(t + 1);
    const float* ap = As + half * LDA + wm * 32 * TM + l31;
    const float* bp = Bs + half * LDB + wn * 32 * TN + l31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = ap[kk * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bp[kk * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 32 * TN + j * 32 + l31;
      if (n >= g.N) continue;
      const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bias;
        if (g.res) v += g.res[m * g.ld_res + n];
        if (g.relu) v = fmaxf(v, 0.f);
        g.out[m * g.ld_out + n] = v;
      }
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int fill_src(SrcDev& d, const tln_gemm_src* s, int64_t M) {
  TLN_REQUIRE(s->d_src && s->cin > 0 && s->ld >= s->cin, "bad gemm source");
  TLN_REQUIRE(s->taps == 1 || s->taps == TLN_TAPS, "taps must be 1 or %d", TLN_TAPS);
  TLN_REQUIRE(s->d_table || s->taps == 1, "taps > 1 needs a table");
  TLN_REQUIRE((s->d_scale == nullptr) == (s->d_shift == nullptr), "scale/shift must come together");
  (void)M;
  d.src = s->d_src;
  d.table = s->d_table;
  d.scale = s->d_scale;
  d.shift = s->d_shift;
  d.src_rows = s->src_rows;
  d.ld = s->ld;
  d.cin = s->cin;
  d.taps = s->taps;
  d.relu = s->relu;
  d.pad = s->pad_value;
  return TLN_OK;
}

template <int TM, int TN, int BK, bool W_NK, bool VEC>
static void launch_gemm(const GemmArgs& g, hipStream_t s) {
  dim3 grid((unsigned)tln_cdiv(g.M, 64 * TM), (unsigned)tln_cdiv(g.N, 64 * TN));
  hipLaunchKernelGGL((k_gather_gemm<TM, TN, BK, W_NK, VEC>), grid, dim3(256), 0, s, g);
}

template <int BK, bool W_NK>
static void dispatch_tiles(const GemmArgs& g, int tm, int tn, hipStream_t s) {
  if (tm == 2 && tn == 2) launch_gemm<2, 2, BK, W_NK, true>(g, s);
  else if (tm == 2) launch_gemm<2, 1, BK, W_NK, true>(g, s);
  else if (tn == 2) launch_gemm<1, 2, BK, W_NK, true>(g, s);
  else launch_gemm<1, 1, BK, W_NK, true>(g, s);
}

// optional tile override for tuning (0 = heuristic)
static int g_force_tm = 0, g_force_tn = 0;
extern "C" void tln_gemm_force_tiles(int tm, int tn) {
  g_force_tm = tm;
  g_force_tn = tn;
}

extern "C" int tln_gather_gemm(int64_t M, int N, const tln_gemm_src* s0, const tln_gemm_src* s1, const float* d_w,
                               int w_is_nk, const float* d_bias, const float* d_residual, int64_t ld_res, int relu,
                               float* d_out, int64_t ld_out, void* stream_) {
  TLN_REQUIRE(s0 && d_w && d_out, "null argument");
  TLN_REQUIRE(M >= 0 && N > 0 && ld_out >= N, "bad gemm shape M=%lld N=%d", (long long)M, N);
  if (M == 0) return TLN_OK;
  TLN_REQUIRE(tln_cdiv(M, 64) < (1ll << 31), "M too large");
  GemmArgs g{};
  g.M = M;
  g.N = N;
  int rc = fill_src(g.s[0], s0, M);
  if (rc) return rc;
  g.K0 = s0->taps * s0->cin;
  g.nsrc = 1;
  int K = g.K0;
  if (s1) {
    rc = fill_src(g.s[1], s1, M);
    if (rc) return rc;
    g.nsrc = 2;
    K += s1->taps * s1->cin;
  }
  g.W = d_w;
  g.ldw = w_is_nk ? K : N;
  g.bias = d_bias;
  g.res = d_residual;
  g.ld_res = ld_res;
  g.relu = relu;
  g.out = d_out;
  g.ld_out = ld_out;
  hipStream_t s = (hipStream_t)stream_;

  bool vec = aligned16(d_w);
  for (int i = 0; i < g.nsrc; ++i) {
    const SrcDev& d = g.s[i];
    vec = vec && aligned16(d.src) && (d.ld % 4 == 0) && (d.cin % 4 == 0);
    if (d.scale) vec = vec && true;
  }
  vec = vec && (w_is_nk ? (K % 4 == 0) : (N % 4 == 0));
  int min_cin = g.s[0].cin;
  if (g.nsrc > 1 && g.s[1].cin < min_cin) min_cin = g.s[1].cin;
  const bool bk32 = (g.s[0].cin % 32 == 0) && (g.nsrc == 1 || g.s[1].cin % 32 == 0);

  if (!vec) {
    if (w_is_nk) launch_gemm<1, 1, 16, true, false>(g, s);
    else launch_gemm<1, 1, 16, false, false>(g, s);
    TLN_LAUNCH_CHECK();
    return TLN_OK;
  }
  int tn = (N > 64) ? 2 : 1, tm = 2;
  auto blocks = [&](int a, int b) { return tln_cdiv(M, 64 * a) * tln_cdiv(N, 64 * b); };
  if (blocks(tm, tn) < 512) tm = 1;
  if (blocks(tm, tn) < 512 && tn == 2) tn = 1;
  if (g_force_tm) tm = g_force_tm;
  if (g_force_tn) tn = g_force_tn;
  if (bk32) {
    if (w_is_nk) dispatch_tiles<32, true>(g, tm, tn, s);
    else dispatch_tiles<32, false>(g, tm, tn, s);
  } else {
    if (w_is_nk) dispatch_tiles<16, true>(g, tm, tn, s);
    else dispatch_tiles<16, false>(g, tm, tn, s);
  }
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// ---------------------------------------------------------------------------------------
// materialised im2row, API parity with Im2RowLattice (lm:301): [M, 9*cin], missing neighbour -> 0
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_im2row(const float* __restrict__ src, int64_t src_rows, int cin,
                                                const int32_t* __restrict__ table, int64_t M,
                                                float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t kw = (int64_t)TLN_TAPS * cin;
  const int64_t m = gid / kw;
  if (m >= M) return;
  const int k = (int)(gid - m * kw);
  const int tap = k / cin, c = k - tap * cin;
  const int srow = table[m * TLN_TAPS + tap];
  out[gid] = (srow >= 0 && srow < src_rows) ? src[(int64_t)srow * cin + c] : 0.f;
}

extern "C" int tln_im2row(const float* d_src, int64_t src_rows, int cin, const int32_t* d_table, int64_t M,
                          float* d_out, void* stream_) {
  TLN_REQUIRE(d_src && d_table && d_out && cin > 0, "null argument");
  if (M <= 0) return TLN_OK;
  const int64_t total = M * TLN_TAPS * cin;
  hipLaunchKernelGGL(k_im2row, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_, d_src,
                     src_rows, cin, d_table, M, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}
