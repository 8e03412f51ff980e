// Argument block shared by the gather-GEMM kernels (gemm.hip: tiled small/medium M and direct; gemm_v2.hip: large M).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SrcDev {
  const float* src;
  const int32_t* table;
  const int32_t* perm;   // optional row order of the product (source 0: rows with equal sets of present taps together)
  const int32_t* order;  // optional launch order of the 128-row blocks of `perm` (heaviest first), gemm_v2 only
  const float* scale;
  const float* shift;
  int64_t src_rows, ld;
  int cin, taps, relu;
  float pad;
  // GroupNorm finalised inside the kernel from per-32-row partial sums (source 0 only)
  const double2* gn_part;
  const float* gn_gamma;
  const float* gn_beta;
  int64_t gn_rows;
  int gn_nblk, gn_groups;
  float gn_eps;
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
  double2* stats;   // optional [cdiv(M,32)][N] (sum, sumsq) of the final values
  float* slab;      // split-K partial tiles [S][tiles][TM*TN*16][GT]
  int* counters;    // split-K arrival counters [tiles], zero between launches
  int splits;
  int fold_taps;    // gemm_v2: taps per partial sum of the split accumulation (fills the padding behind `splits`)
  unsigned long long* dbg;  // diagnostic: s_memtime stamps of block (0,0,0) (tools/gemm_stamps.py), else NULL
  // the GRU cell as ONE product (gemm_v2.hip, k_gather_gemm_v2_gru): weights and bias of the second source (h)
  const float* W2;
  const float* bias2;
};


#define TLN_GEMM_MULTI_MAX 8
// up to TLN_GEMM_MULTI_MAX products of one shape class in one launch (blockIdx.z = product)
template <int NP>
struct GemmArgsN {
  GemmArgs a[NP];
  int xcd = 0;   // gemm_v2's shared launches: products dealt to the XCDs (v2_multi_block)
};

// lattice.hip: the row order that belongs to a tap table, if it was built for exactly `rows` rows
const int32_t* tln_table_perm(const int32_t* table, int64_t rows);
const int32_t* tln_table_tile_order(const int32_t* table, int64_t rows);

// gemm_v2.hip: the large-M kernel (block tile 128 x N, operands staged by LDS-DMA).  tln_gemm_v2_ok decides from the
// prepared arguments alone, so every route (operator call, frame program, lock-step group) takes the same kernel.
// (every decision takes the caller's tln_options explicitly: there is no file-scope switch)
bool tln_gemm_v2_ok(const GemmArgs& g, bool w_is_nk, bool vec, const tln_options& o);
int tln_gemm_v2_launch(GemmArgs& g, bool w_is_nk, hipStream_t s, const tln_options& o);
// n products of one shape class whose rows TOGETHER make a large M (lock-stepped sequences on a coarse level): one launch
bool tln_gemm_v2_multi_ok(const GemmArgs* g, int n, bool w_is_nk, const bool* vec, const tln_options& o);
int tln_gemm_v2_launch_multi(GemmArgs* g, int n, bool w_is_nk, hipStream_t s, const tln_options& o);
// the GRU cell as one two-source product with the gates in its epilogue (large V, C a multiple of 64)
bool tln_gemm_v2_gru_ok(int64_t V, int64_t Vh, int C, const tln_options& o);
int tln_gemm_v2_launch_gru(const float* d_x, const float* d_h, int64_t Vh, int64_t V, int C, const float* d_w_ih,
                           const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out, hipStream_t s);
int tln_gemm_v2_launch_gru_multi(int n, const float* const* d_x, const float* const* d_h, const int64_t* Vh, const int64_t* V,
                                 int C, const float* d_w_ih, const float* d_w_hh, const float* d_b_ih, const float* d_b_hh,
                                 float* const* d_out, hipStream_t s);
