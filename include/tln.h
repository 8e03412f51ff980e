/* tln.h — C ABI of libtln_hip.so: the MI355X (gfx950) implementation of the lattice
 * operators temporal_latticenet calls through the un-vendored `latticenet` pybind module
 * and `latticenet_py` package (reference README.md:47; call sites cited per entry point).
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory owned by the caller (PyTorch); the library
 *     never frees it and never keeps it past the call.  Table memory is owned by the handle.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it.
 *   - return value: 0 = ok, <0 = TLN_E_* ; nothing throws across the boundary.
 *   - fp32 everywhere, int32 indices, row-major [rows, channels].
 *   - pos_dim is 3 (the reference's AFlow hard-codes 9 = 2(d+1)+1 taps, lattice_modules.py:310).
 */
#ifndef TLN_H
#define TLN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TLN_OK 0
#define TLN_E_INVALID (-1)   /* bad argument / unsupported shape            */
#define TLN_E_HIP (-2)       /* a HIP runtime call failed (see tln_last_error) */
#define TLN_E_STATE (-3)     /* call order violated (e.g. pool without CSR)  */
#define TLN_E_CAPACITY (-4)  /* workspace / table capacity exceeded          */

#define TLN_TAPS 9           /* 2(d+1) one-hop neighbours + centre (LAST)    */

typedef struct tln_lattice tln_lattice_t;

const char* tln_last_error(void);
int tln_version(void);

/* ---- Lattice handle: replaces latticenet.Lattice (train_ln.py:106, 239) -------------- */
int tln_lattice_create(tln_lattice_t** out, int pos_dim, const double* sigmas, int64_t capacity);
int tln_lattice_destroy(tln_lattice_t* l);
/* reset_hashmap=True of DistributeLatticeModule (models.py:287-298): clears every level */
int tln_lattice_clear(tln_lattice_t* l, void* stream);
int64_t tln_lattice_nr_vertices(const tln_lattice_t* l);      /* Lattice.nr_lattice_vertices(), train_ln.py:220 */
int64_t tln_lattice_capacity(const tln_lattice_t* l);
int tln_lattice_level(const tln_lattice_t* l);
int64_t tln_lattice_overflow_rows(const tln_lattice_t* l);    /* rows that got index -1 in the last distribute */
/* vertex keys [V,3] int32 (first d coordinates), for tests and the multi-GPU key exchange */
int tln_lattice_keys(const tln_lattice_t* l, int32_t* d_keys_out, int64_t max_rows, void* stream);
/* insert externally supplied keys (row order = first-touch order); used by the frame-sharded path */
int tln_lattice_insert_keys(tln_lattice_t* l, const int32_t* d_keys, int64_t n, int32_t* d_indices_out, void* stream);

/* ---- K1 distribute: DistributeLatticeModule.forward (models.py:298) ------------------- */
/* d_positions [n,3], d_values [n,val_dim] -> d_distributed [4n, 3+val_dim+1],
 * d_indices [4n] (-1 = not inserted), d_weights [4n].  subtract_mean: rows carry
 * position - mean position of their vertex (0 for the *_no_local_mean experiments).
 * Leaves a vertex-sorted row list (CSR) in the handle for tln_pointnet_pool. */
int tln_distribute(tln_lattice_t* l, const float* d_positions, const float* d_values, int64_t n,
                   int val_dim, int subtract_mean, float* d_distributed, int32_t* d_indices,
                   float* d_weights, void* stream);

/* rebuild the CSR from caller-supplied indices (R rows, -1 folded into the tail bucket) */
int tln_build_csr(tln_lattice_t* l, const int32_t* d_indices, int64_t rows, void* stream);

/* copy the handle's CSR out (any pointer may be NULL): order[rows] = row ids sorted stably by vertex,
 * sorted_vertex[rows] (rejected rows carry V), seg_start[V+2]; *rows_out = rows of the last build (0 = none).
 * The reference keeps no such structure (torch_scatter works on the unsorted rows, lm:485); test/debug surface. */
int tln_lattice_csr(tln_lattice_t* l, int32_t* d_order, int32_t* d_sorted_vertex, int32_t* d_seg_start,
                    int64_t* rows_out, void* stream);

/* ---- K2 PointNet pool: PointNetSeqModule.forward lm:448-530 --------------------------- */
/* per-row MLP (nr_layers linears, ReLU between) on distributed[:, :cin] then segment-max by
 * vertex with argmax, barycentric-of-argmax (with the lm:514 clamp quirk), <min_points mask.
 * d_w[i] is torch Linear weight [cout_i, cin_i], d_b[i] bias.  nr_layers == 0 => no MLP.
 * out [V, 2*cout_last]. */
int tln_pointnet_pool(tln_lattice_t* l, const float* d_distributed, int64_t rows, int dist_cols,
                      int nr_layers, const float* const* d_w, const float* const* d_b,
                      const int* dims /* [nr_layers+1] */, int min_points, float* d_out, void* stream);

/* same, and d_argrow [V, cout_last] int32 = the row whose MLP output is the pooled value (-1: empty / masked
 * vertex) — what the backward pass of the pool needs */
int tln_pointnet_pool_ex(tln_lattice_t* l, const float* d_distributed, int64_t rows, int dist_cols,
                         int nr_layers, const float* const* d_w, const float* const* d_b,
                         const int* dims, int min_points, float* d_out, int32_t* d_argrow, void* stream);

/* ---- structure: neighbour tables, coarse level ---------------------------------------- */
/* [V,9] table of the level itself (centre last = own index); cached until the level grows.
 * Replaces the hashing inside Im2RowLattice / Im2RowIndicesLattice (lm:301, 304). */
int tln_neighbour_table(tln_lattice_t* l, const int32_t** d_table_out, void* stream);
/* GnReluCoarsen / create-coarse-verts (models.py:353): returns the persistent child level,
 * extended (append-only) by the fine vertices added since the last call. */
int tln_coarsen(tln_lattice_t* fine, tln_lattice_t** coarse_out, void* stream);
/* once per frame, after tln_distribute: extends the coarse levels and (re)builds every stale neighbour / cross-level
 * table of the stack in one launch; afterwards the table getters return cached pointers */
int tln_lattice_prepare_levels(tln_lattice_t* level0, int nr_coarse_levels, void* stream);
/* [V_coarse,9] rows into the fine level (coarsen conv) / [V_fine,9] rows into the coarse level (finefy) */
int tln_coarse_to_fine_table(tln_lattice_t* coarse, const int32_t** d_table_out, void* stream);
int tln_fine_to_coarse_table(tln_lattice_t* coarse, const int32_t** d_table_out, void* stream);

/* ---- K3+K4 gather-GEMM (implicit im2row): ConvLatticeModule, Coarsen, Finefy, 1x1 ------ */
typedef struct {
  const float* d_src;      /* [src_rows, cin] source rows                                  */
  int64_t src_rows;        /* rows >= src_rows read as pad_value (hidden-state padding)    */
  int64_t ld;              /* row stride of d_src in floats                                */
  int cin;                 /* channels per tap                                             */
  int taps;                /* 1 (identity / 1x1) or TLN_TAPS                               */
  const int32_t* d_table;  /* [M,taps] row ids, -1 => zero row; NULL => row m itself       */
  float pad_value;
  const float* d_scale;    /* optional per-channel prologue a*scale+shift (GroupNorm apply) */
  const float* d_shift;
  int relu;                /* ReLU after the affine prologue                               */
  /* GroupNorm of this source finalised INSIDE the GEMM (source 0 only): per-32-row (sum,sumsq) doubles
   * [ceil(gn_rows/32)][cin] as written by tln_gather_gemm_ex / tln_groupnorm_partials.  When set, d_scale/d_shift
   * are only scratch [cin] each for the large-V fallback. */
  const void* d_gn_partials;
  const float* d_gn_gamma;
  const float* d_gn_beta;
  int64_t gn_rows;
  int gn_groups;
  float gn_eps;
} tln_gemm_src;

/* C[M,N] = epilogue( concat_k(src0, src1) @ W ), K = taps0*cin0 (+ taps1*cin1).
 * W is [K,N] row-major (w_is_nk=0, ConvLatticeModule layout lm:291) or [N,K] (w_is_nk=1, torch Linear).
 * epilogue: + bias[N], + residual[M,N], ReLU. */
int tln_gather_gemm(int64_t M, int N, const tln_gemm_src* s0, const tln_gemm_src* s1,
                    const float* d_w, int w_is_nk, const float* d_bias, const float* d_residual,
                    int64_t ld_res, int relu, float* d_out, int64_t ld_out, void* stream);

/* same, plus the GroupNorm statistics of the output for the NEXT layer: d_stats (optional) receives
 * [ceil(M/32)][N] pairs of doubles (sum, sum of squares) over each 32-row block of the final values, i.e. the
 * input format of tln_groupnorm_from_partials */
int tln_gather_gemm_ex(int64_t M, int N, const tln_gemm_src* s0, const tln_gemm_src* s1,
                       const float* d_w, int w_is_nk, const float* d_bias, const float* d_residual,
                       int64_t ld_res, int relu, float* d_out, int64_t ld_out, void* d_stats, void* stream);

/* tuning hook: force the block tile (TM,TN in {1,2}; 0 = heuristic) */
void tln_gemm_force_tiles(int tm, int tn);
/* tuning hook: force the number of K-groups per block (1, 2, 4; 0 = heuristic) for 64x64 tiles */
void tln_gemm_force_groups(int groups);
/* tuning hook: force the split-K slices over the grid and the tile height (wm: 1 = 32 rows, 2 = 64 rows) */
void tln_gemm_force_splits(int splits, int wm);
/* tuning hook: the small-M "direct" kernel (operands from global memory, no LDS tiles): 0 = heuristic,
 * 1 = whenever the shape is eligible (channels multiple of 32, aligned), -1 = never */
void tln_gemm_force_direct(int mode);
/* diagnostic hook: block (0,0,0) of every following gather-GEMM writes five s_memtime stamps (start, after the
 * index/GroupNorm prologue, after the K loop, after the reductions, end) to d_buf (5 x u64); NULL switches it off */
void tln_gemm_debug_stamps(void* d_buf);

/* materialised im2row (API parity with Im2RowLattice / Im2RowIndicesLattice, lm:301-304) */
int tln_im2row(const float* d_src, int64_t src_rows, int cin, const int32_t* d_table, int64_t M,
               float* d_out /* [M, 9*cin] */, void* stream);

/* ---- K7 GroupNorm statistics over all vertices (Gn / GnRelu* modules) ------------------ */
/* writes per-channel scale/shift so that y = x*scale + shift == GroupNorm(x)*gamma + beta */
int64_t tln_groupnorm_ws_bytes(int64_t V, int C);
int tln_groupnorm_stats(const float* d_x, int64_t V, int C, int groups, const float* d_gamma,
                        const float* d_beta, float eps, float* d_scale, float* d_shift, void* d_ws,
                        int64_t ws_bytes, void* stream);
/* second half of tln_groupnorm_stats on partial sums that already exist (written by tln_gather_gemm_ex):
 * d_partials = [ceil(V/32)][C] (sum, sumsq) doubles */
/* first half only: per-32-row partial sums of x [V,C] -> d_partials [ceil(V/32)][C] pairs of doubles */
int tln_groupnorm_partials(const float* d_x, int64_t V, int C, void* d_partials, void* stream);
int tln_groupnorm_from_partials(const void* d_partials, int64_t V, int C, int groups, const float* d_gamma,
                                const float* d_beta, float eps, float* d_scale, float* d_shift, void* stream);
/* GroupNorm (+ReLU) folded into a gather-GEMM: statistics -> per-channel scale/shift -> tln_gather_gemm_ex in ONE
 * host call (GnRelu1x1 / GnReluConv / GnReluCoarsen / GnReluFinefy of the reference's operator package) */
typedef struct {
  const void* d_partials;   /* [ceil(V/32)][C] (sum,sumsq) doubles from the producer's epilogue, or NULL      */
  const float* d_x;         /* [V,C] the normalised tensor (read only when d_partials is NULL)                */
  int64_t V;
  int C, groups, relu;
  const float* d_gamma;
  const float* d_beta;
  float eps;
  float* d_scale_shift;     /* [2,C] out: scale then shift                                                    */
  void* d_ws;               /* tln_groupnorm_ws_bytes(V,C) bytes, only when d_partials is NULL                */
  int64_t ws_bytes;
} tln_gn_desc;
int tln_gn_gather_gemm(const tln_gn_desc* gn, int64_t M, int N, const tln_gemm_src* s0, const tln_gemm_src* s1,
                       const float* d_w, int w_is_nk, const float* d_bias, const float* d_residual,
                       int64_t ld_res, int relu, float* d_out, int64_t ld_out, void* d_stats, void* stream);
int tln_affine_act(const float* d_x, int64_t V, int C, const float* d_scale, const float* d_shift,
                   int relu, float* d_out, void* stream);

/* ---- K9 GRU fusion: GRUModule.forward lm:53-66 ----------------------------------------- */
/* x [V,C]; h [Vh,C] (already through hidden_linear), rows >= Vh are zero; GRUCell weights
 * w_ih,w_hh [3C,C], b_ih,b_hh [3C]; out [V,C]. */
int tln_gru_cell(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const float* d_w_ih,
                 const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out,
                 float* d_ws /* [V,6C] */, int64_t ws_floats, void* stream);

/* ---- K10 AFlow correlation: CustomKernelConvLatticeIm2RowModule.forward lm:282-339 ------ */
/* x [V,C] current features, h [Vh,C] previous hidden state (rows >= Vh = pad_value -999999),
 * table [V,9]; out [V,C] (+bias), weights [V,9], nbr_idx [V,9]. */
int tln_aflow(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const int32_t* d_table,
              float alpha, float beta, float pad_value, int use_center, const float* d_bias,
              float* d_out, float* d_weights, int32_t* d_nbr_idx, void* stream);

/* ---- K8 slice: SliceFastCUDALatticeModule (models.py:465) / SliceLatticeModule ---------- */
/* gather for the delta-weight head: out [n, 4*(cb+1)] = for r: [w_r * b[idx_r, :cb], w_r] */
int tln_slice_gather(const float* d_lv, int64_t V, int cb, const int32_t* d_indices, const float* d_weights,
                     int64_t n, float* d_out, void* stream);
/* out [n,C] = sum_r (w_r + dw_r) * lv[idx_r]   (d_delta may be NULL) */
int tln_slice(const float* d_lv, int64_t V, int C, const int32_t* d_indices, const float* d_weights,
              const float* d_delta, int64_t n, float* d_out, void* stream);

/* ---- K11 plain splat (SplatLatticeModule) ---------------------------------------------- */
/* out [V, val_dim+1] = sum over rows of w * [values, 1]  (uses the CSR of the last distribute) */
int tln_splat(tln_lattice_t* l, const float* d_values, int val_dim, const float* d_weights, int64_t rows,
              float* d_out, void* stream);

/* ---- torch_scatter 2.0.4 equivalents used at lm:485-520, models.py:454 ------------------ */
int tln_scatter_max(const float* d_src, const int64_t* d_index, int64_t rows, int C, int64_t out_rows,
                    float* d_out, int64_t* d_argmax, void* d_ws /* out_rows*C*8 bytes */, int64_t ws_bytes,
                    void* stream);
int tln_scatter_add(const float* d_src, const int64_t* d_index, int64_t rows, int C, int64_t out_rows,
                    float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
