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

/* ---- Kernel-selection options -------------------------------------------------------------
 * The library holds NO process-wide mutable state ("no global state except explicit handles"): every
 * test / measurement switch is a field of this struct, and the struct travels explicitly -- stored in a lattice handle
 * (tln_lattice_set_options: K1, K2), in a frame program (tln_program_set_options: every product, GRU cell and batched
 * stage the program issues) or passed with an operator call (tln_gather_gemm_opt, tln_gather_gemm_multi_opt,
 * tln_gru_cell_opt).  NULL / tln_options_init() = what the library chooses by itself; results are the same bits under
 * every setting unless a field says otherwise.  (Until round 3 these were tln_*_config / tln_gemm_force_* setters on
 * file-scope variables, racy under the four host threads of streams.py.) */
typedef struct tln_options {
  int k1_legacy;         /* K1: 0 = partitioned kernels (default), 1 = one global atomic per row (always taken for val_dim > 1) */
  int k1_bucket_rows;    /* K1: rows per bucket the partitioned kernels aim at; 0 = default (512) */
  int pool_mode;         /* K2: -1 = default, 0 = VALU fma chains, 1 / 2 = wide layers on the matrix cores, 3 = VALU, two rows per
                            step (all the same bits) */
  int gemm_direct;       /* small-M "direct" kernel: 0 = heuristic, 1 = whenever eligible, -1 = never */
  int gemm_pair_off;     /* 1 = tln_gather_gemm_pair / _multi as separate launches */
  int gemm_tn;           /* force the column tile of the tiled kernel (1 / 2: 64 / 128 columns; 0 = heuristic) */
  int gemm_groups;       /* force the K-groups (waves per tile; 0 = heuristic): changes the K-summation order */
  int gemm_splits;       /* force the split-K slices over the grid (0 = heuristic) */
  int gemm_wm;           /* force the tile height with gemm_splits (1 = 32 rows, 2 = 64 rows; 0 = heuristic) */
  int v2_off;            /* large-M kernel (gemm_v2.hip): bit 0 off, bit 2 no row orders / tap skipping, bit 3 = 64-row
                          * tiles for every shared 128-column launch with the GroupNorm prologue, bit 4 = never */
  int64_t v2_min_m;      /* smallest M that takes the large-M kernel; 0 = default (12288, env TLN_V2_MIN_M) */
  void* gemm_stamps;     /* diagnostic: block (0,0,0) of a gather-GEMM writes five s_memtime stamps to this device buffer */
  int group_off_mask;    /* frame programs in group mode: bit k = ops of kind k (TLN_OP_*) launched per program instead
                          * of batched; bit 0 = the K1 / coarse-level / table batches */
} tln_options;
void tln_options_init(tln_options* o);    /* the defaults (pool_mode = -1, everything else 0) */

/* ---- Lattice handle: replaces latticenet.Lattice (train_ln.py:106, 239) -------------- */
int tln_lattice_create(tln_lattice_t** out, int pos_dim, const double* sigmas, int64_t capacity);
/* the same with the lattice scale constant c of scale_i = c / (sigma_i * sqrt((i+1)(i+2))) given: a free choice of the
 * un-vendored lattice_net dependency (README.md:47; cfg key lattice_gpu.scale_constant here).  0 = the default,
 * Adams 2010's (d+1)*sqrt(2/3) (meets the sizing hint of seq_config/lnn_train_semantic_kitti.cfg:71); 1.0 = the factor
 * dropped (what a checkpoint trained against such a build expects).  Coarse levels inherit it. */
int tln_lattice_create_ex(tln_lattice_t** out, int pos_dim, const double* sigmas, int64_t capacity,
                          double scale_constant);
double tln_lattice_default_scale_constant(void);
double tln_lattice_scale_constant(const tln_lattice_t* l);
int tln_lattice_destroy(tln_lattice_t* l);
/* copies *opt (NULL: the defaults) into the handle: k1_* for its distributes, pool_mode for its pools.  A batch of
 * lattices (tln_distribute_begin_multi, tln_pointnet_pool_multi) follows the first lattice's options. */
int tln_lattice_set_options(tln_lattice_t* l, const tln_options* opt);
/* reset_hashmap=True of DistributeLatticeModule (models.py:287-298): clears every level */
int tln_lattice_clear(tln_lattice_t* l, void* stream);
/* the same for n lattices (lock-stepped sequences), as few launches as their levels allow */
int tln_lattice_clear_multi(tln_lattice_t* const* l, int n, void* stream);
int64_t tln_lattice_nr_vertices(const tln_lattice_t* l);      /* Lattice.nr_lattice_vertices(), train_ln.py:220 */
int64_t tln_lattice_capacity(const tln_lattice_t* l);
int tln_lattice_level(const tln_lattice_t* l);
/* device memory the handle owns over its whole level stack, in bytes: out[0] hash tables + per-vertex arrays, [1] per-row
 * workspaces of the distribute (records, vertex bins, sort scratch), [2] the pool's accumulators, [3] neighbour /
 * cross-level tables + row orders, [4] total */
int tln_lattice_memory(const tln_lattice_t* l, int64_t* out /* [5] */);
int64_t tln_lattice_overflow_rows(const tln_lattice_t* l);    /* rows that got index -1 in the last distribute */
/* vertex keys [V,3] int32 (first d coordinates), for tests and the multi-GPU key exchange */
int tln_lattice_keys(const tln_lattice_t* l, int32_t* d_keys_out, int64_t max_rows, void* stream);
/* insert externally supplied keys (row order = first-touch order); used by the frame-sharded path */
int tln_lattice_insert_keys(tln_lattice_t* l, const int32_t* d_keys, int64_t n, int32_t* d_indices_out, void* stream);

/* ---- K1 distribute: DistributeLatticeModule.forward (models.py:298) ------------------- */
/* d_positions [n,3], d_values [n,val_dim] -> d_distributed [4n, 3+val_dim+1],
 * d_indices [4n] (-1 = not inserted), d_weights [4n].  subtract_mean: rows carry
 * position - mean position of their vertex (0 for the *_no_local_mean experiments).
 * Leaves the frame's rows grouped by vertex ("bins": payload moved to one contiguous segment per vertex, no sort) in
 * the handle; tln_pointnet_pool called with the same d_distributed / rows reads them.  d_distributed may be NULL when
 * only the pool of the same frame consumes the rows (the frame program does that). */
int tln_distribute(tln_lattice_t* l, const float* d_positions, const float* d_values, int64_t n,
                   int val_dim, int subtract_mean, float* d_distributed, int32_t* d_indices,
                   float* d_weights, void* stream);
/* the same in two halves: _begin enqueues the insertion / numbering and starts the fetch of the vertex counters,
 * and the bins, _finish waits for that fetch only and enqueues the mean subtraction of d_distributed.  A caller with several lattices
 * on one stream (lock-step groups) begins all of them before it finishes the first: one wait instead of one per
 * lattice.  The buffers given to _begin must stay valid until _finish has returned. */
int tln_distribute_begin(tln_lattice_t* l, const float* d_positions, const float* d_values, int64_t n,
                   int val_dim, int subtract_mean, float* d_distributed, int32_t* d_indices,
                   float* d_weights, void* stream);
int tln_distribute_finish(tln_lattice_t* l, void* stream);
/* the first halves of the distributes of n lock-stepped sequences (n different lattices, one stream) as ONE batch: the
 * four K1 kernels take the frames of up to eight lattices in one launch each (blockIdx.y = frame).  Every lattice is
 * then finished by its own tln_distribute_finish.  Same results as n tln_distribute_begin calls, bit for bit. */
typedef struct {
  tln_lattice_t* l;
  const float* d_positions;
  const float* d_values;
  int64_t n;
  int val_dim;
  int subtract_mean;
  float* d_distributed;
  int32_t* d_indices;
  float* d_weights;
} tln_distribute_call;
int tln_distribute_begin_multi(const tln_distribute_call* calls, int n, void* stream);

/* forget the bins of the last distribute (the caller edited d_distributed in place): the next pool goes through a CSR */
int tln_lattice_drop_bins(tln_lattice_t* l);
/* (K1 variant and bucket size: tln_options.k1_legacy / k1_bucket_rows via tln_lattice_set_options.)  A bucket (one
 * workgroup, a 1024-entry LDS table) that meets more distinct keys than its table holds makes the library redo the frame
 * with the per-row-atomic kernels (same results); tln_lattice_bucket_fallbacks counts those frames. */
int64_t tln_lattice_bucket_fallbacks(const tln_lattice_t* l);

/* build the vertex-sorted row list (CSR) from caller-supplied indices (R rows, -1 folded into the tail bucket) */
int tln_build_csr(tln_lattice_t* l, const int32_t* d_indices, int64_t rows, void* stream);

/* copy the handle's CSR out (any pointer may be NULL): order[rows] = row ids sorted stably by vertex,
 * sorted_vertex[rows] (rejected rows carry V), seg_start[V+2]; *rows_out = rows of the last build (0 = none).
 * The reference keeps no such structure (torch_scatter works on the unsorted rows, lm:485); test/debug surface. */
int tln_lattice_csr(tln_lattice_t* l, int32_t* d_order, int32_t* d_sorted_vertex, int32_t* d_seg_start,
                    int64_t* rows_out, void* stream);

/* ---- K2 PointNet pool: PointNetSeqModule.forward lm:448-530 --------------------------- */
/* (Kernel choice for the 4-16-32-64 PointNet MLP pooled from the bins of the frame's distribute: tln_options.pool_mode of
 * the lattice handle; identical bits under every mode.) */

/* per-row MLP (nr_layers linears, ReLU between) on distributed[:, :cin] then segment-max by
 * vertex with argmax, barycentric-of-argmax (with the lm:514 clamp quirk), <min_points mask.
 * d_w[i] is torch Linear weight [cout_i, cin_i], d_b[i] bias.  nr_layers == 0 => no MLP.
 * out [V, 2*cout_last]. */
int tln_pointnet_pool(tln_lattice_t* l, const float* d_distributed, int64_t rows, int dist_cols,
                      int nr_layers, const float* const* d_w, const float* const* d_b,
                      const int* dims /* [nr_layers+1] */, int min_points, float* d_out, void* stream);

/* the pools of n lock-stepped sequences (same MLP; every lattice's own bins / rows / output) with one launch per
 * kernel (blockIdx.y = sequence).  Same results as n tln_pointnet_pool calls. */
typedef struct {
  tln_lattice_t* l;
  const float* d_distributed;
  int64_t rows;
  float* d_out;
} tln_pool_call;
int tln_pointnet_pool_multi(const tln_pool_call* calls, int n, int dist_cols, int nr_layers, const float* const* d_w,
                            const float* const* d_b, const int* dims, int min_points, void* stream);

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
/* the same in two halves, so that work that only needs level 0 can be launched while the coarse levels' vertex
 * counts are still on their way to the host: _begin extends every coarse level, starts ONE asynchronous fetch of
 * their counters and builds the level-0 neighbour table; v_bound_out[0] = V0 (exact), v_bound_out[i] >= the new vertex
 * count of level i (may be NULL).  _finish waits for that fetch only (an event, not the stream), publishes the exact
 * counts and builds the coarse tables.  Every entry point that looks at a coarse level finishes a pending fetch. */
int tln_lattice_prepare_levels_begin(tln_lattice_t* l, int nr_coarse_levels, int64_t* v_bound_out, void* stream);
int tln_lattice_prepare_levels_finish(tln_lattice_t* l, void* stream);
/* both halves for the level stacks of n lock-stepped sequences (n level-0 lattices): every launch carries the work of
 * all of them (blockIdx.y / job list = lattice).  v_bound_out is [n][TLN_MAX_LEVELS] (or NULL). */
int tln_lattice_prepare_levels_begin_multi(tln_lattice_t* const* l, int n, int nr_coarse_levels, int64_t* v_bound_out,
                                           void* stream);
int tln_lattice_prepare_levels_finish_multi(tln_lattice_t* const* l, int n, void* stream);
/* the coarse level of `l` as it stands (NULL if none yet); no side effects */
tln_lattice_t* tln_lattice_coarse_level(tln_lattice_t* l);
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

/* Two independent products in one launch: for a host that steps two sequences in lock-step on one stream (the
 * frame program's pair mode, tln_program_run_pair).  Each call is what tln_gather_gemm_ex would take.  When both
 * are products of the same shape class (same N, K layout, channel counts, weight layout; M may differ) that take the
 * small-M kernel, they run as ONE launch (blockIdx.z = problem) and each result is what its own launch would give up
 * to the order of the K summation (the waves per tile are chosen for the pair); otherwise as two launches. */
typedef struct {
  int64_t M;
  int N;
  const tln_gemm_src* s0;
  const tln_gemm_src* s1;   /* NULL: one source */
  const float* d_w;
  int w_is_nk;
  const float* d_bias;
  const float* d_residual;
  int64_t ld_res;
  int relu;
  float* d_out;
  int64_t ld_out;
  void* d_stats;
} tln_gemm_call;
int tln_gather_gemm_pair(const tln_gemm_call* a, const tln_gemm_call* b, void* stream);
/* the same for n = 1..8 calls (more than 8: one launch each) */
int tln_gather_gemm_multi(const tln_gemm_call* calls, int n, void* stream);
/* the same under explicit kernel-selection options (NULL = defaults); tln_gather_gemm_opt = one call */
int tln_gather_gemm_opt(const tln_gemm_call* call, const tln_options* opt, void* stream);
int tln_gather_gemm_multi_opt(const tln_gemm_call* calls, int n, const tln_options* opt, void* stream);

/* ---- backward of the gather-GEMM (training: train_ln.py:212-233 calls loss.backward() through these products) ------
 * dW [taps*cin, N] (the [K, N] layout of lm:291; a Linear's [N, K] gradient is its transpose) =
 * sum_m gather(src, table)[m, :]^T dout[m, :].  MFMA tiles over (tap, 32 channels, 32-64 columns), M cut into slices
 * whose partial tiles are added in slice order: deterministic, no atomics.  taps 1 (d_table NULL:
 * row m itself) or TLN_TAPS.  d_ws: tln_gather_gemm_dw_ws_floats(M, cin, taps, N) floats.
 * dA needs no entry point of its own: for a level's own neighbour table (paired taps) and for 1x1 products it is a
 * forward gather-GEMM with rearranged weights (temporal_latticenet_amd/autograd.py). */
int64_t tln_gather_gemm_dw_ws_floats(int64_t M, int cin, int taps, int N);
int tln_gather_gemm_dw(const float* d_src, int64_t src_rows, int cin, const int32_t* d_table, int taps,
                       const float* d_dout, int64_t M, int N, float* d_dw, float* d_ws, int64_t ws_floats, void* stream);
/* backward of the slice blends as segment sums over the lattice's vertex-sorted row list (tln_build_csr over the same
 * rows): d_out [V, C] = sum over the rows of v of (w_row + delta_row) * dvals[row >> 2 (per_row: row)][:C]; and
 * d_dw [rows] = dot(lv[idx_row], dout[row >> 2]).  Deterministic (no float atomics). */
int tln_slice_blend_bwd_lv(tln_lattice_t* l, const float* d_dvals, int64_t ld, int C, int per_row, const float* d_weights,
                           const float* d_delta, int64_t rows, float* d_out, void* stream);
int tln_slice_blend_bwd_w(const float* d_lv, int64_t V, int C, const int32_t* d_indices, const float* d_dout,
                          int64_t rows, float* d_dw, void* stream);

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
/* the first half for n tensors of one width (the lock-stepped sequences of a stream) in one launch */
typedef struct {
  const float* d_x;
  int64_t V;
  void* d_partials;
} tln_gn_partials_call;
int tln_groupnorm_partials_multi(const tln_gn_partials_call* calls, int n, int C, void* stream);
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
int tln_gn_gather_gemm_opt(const tln_gn_desc* gn, int64_t M, int N, const tln_gemm_src* s0, const tln_gemm_src* s1,
                           const float* d_w, int w_is_nk, const float* d_bias, const float* d_residual,
                           int64_t ld_res, int relu, float* d_out, int64_t ld_out, void* d_stats,
                           const tln_options* opt, void* stream);
int tln_affine_act(const float* d_x, int64_t V, int C, const float* d_scale, const float* d_shift,
                   int relu, float* d_out, void* stream);

/* ---- K9 GRU fusion: GRUModule.forward lm:53-66 ----------------------------------------- */
/* x [V,C]; h [Vh,C] (already through hidden_linear), rows >= Vh are zero; GRUCell weights
 * w_ih,w_hh [3C,C], b_ih,b_hh [3C]; out [V,C]. */
int tln_gru_cell(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const float* d_w_ih,
                 const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out,
                 float* d_ws /* [V,6C] */, int64_t ws_floats, void* stream);

/* the cells of n lock-stepped sequences (same GRUCell weights; every call's own x / h / out / workspace) */
typedef struct {
  const float* d_x;
  const float* d_h;
  int64_t V, Vh;
  float* d_out;
  float* d_ws;
  int64_t ws_floats;
} tln_gru_call;
int tln_gru_cell_multi(const tln_gru_call* calls, int n, int C, const float* d_w_ih, const float* d_w_hh,
                       const float* d_b_ih, const float* d_b_hh, void* stream);
/* both under explicit kernel-selection options (NULL = defaults: the entry points above) */
int tln_gru_cell_opt(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const float* d_w_ih,
                     const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out,
                     float* d_ws /* [V,6C] */, int64_t ws_floats, const tln_options* opt, void* stream);
int tln_gru_cell_multi_opt(const tln_gru_call* calls, int n, int C, const float* d_w_ih, const float* d_w_hh,
                           const float* d_b_ih, const float* d_b_hh, const tln_options* opt, void* stream);

/* ---- element-wise steps of the alternative fusion modules (rnn_modules = lstm / maxpool / cga, lm:17-185) ---- */
/* LSTMModule lm:36-38: gates [V,4C] in torch LSTMCell order i|f|g|o, zero cell state: out = sig(o)*tanh(sig(i)*tanh(g)) */
int tln_lstm_gates(const float* d_gates, int64_t V, int C, float* d_out, void* stream);
/* TemporalMaxPoolModule lm:138-141: out [V,C] = max(h padded to V rows with pad_value, x) */
int tln_temporal_max(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, float pad_value,
                     float* d_out, void* stream);
/* CrossframeGlobalAttentionModule lm:104-112: out = sigmoid(a * scale) * x for rows < Vh, x for the rows born in this
 * frame (their gate is set to 1, lm:109-110) */
int tln_cga_gate(const float* d_a, const float* d_x, int64_t V, int64_t Vh, int C, float scale, float* d_out,
                 void* stream);
/* PointNetSeqModule lm:555-562 (early max-pool fusion): rows whose first `half` channels are all zero -> `value` */
int tln_fill_empty_rows(const float* d_x, int64_t V, int C, int half, float value, float* d_out, void* stream);

/* ---- K10 AFlow correlation: CustomKernelConvLatticeIm2RowModule.forward lm:282-339 ------ */
/* x [V,C] current features, h [Vh,C] previous hidden state (rows >= Vh = pad_value -999999),
 * table [V,9]; out [V,C] (+bias), weights [V,9], nbr_idx [V,9]. */
int tln_aflow(const float* d_x, const float* d_h, int64_t V, int64_t Vh, int C, const int32_t* d_table,
              float alpha, float beta, float pad_value, int use_center, const float* d_bias,
              float* d_out, float* d_weights, int32_t* d_nbr_idx, void* stream);

/* the correlations of n lock-stepped sequences (same alpha / beta / bias; every call's own tensors) in one launch */
typedef struct {
  const float* d_x;
  const float* d_h;
  int64_t V, Vh;
  const int32_t* d_table;
  float* d_out;
  float* d_weights;
  int32_t* d_nbr_idx;
} tln_aflow_call;
int tln_aflow_multi(const tln_aflow_call* calls, int n, int C, float alpha, float beta, float pad_value, int use_center,
                    const float* d_bias, void* stream);

/* ---- K8 slice: SliceFastCUDALatticeModule (models.py:465) / SliceLatticeModule ---------- */
/* gather for the delta-weight head: out [n, 4*(cb+1)] = for r: [w_r * b[idx_r, :cb], w_r] */
int tln_slice_gather(const float* d_lv, int64_t V, int cb, const int32_t* d_indices, const float* d_weights,
                     int64_t n, float* d_out, void* stream);
/* out [n,C] = sum_r (w_r + dw_r) * lv[idx_r] + bias   (d_delta, d_bias [C] may be NULL) */
int tln_slice(const float* d_lv, int64_t V, int C, const int32_t* d_indices, const float* d_weights,
              const float* d_delta, const float* d_bias, int64_t n, float* d_out, void* stream);

/* the whole DeformSlice head of SliceFastCUDALatticeModule (models.py:465) per point in one kernel:
 * g = [w_r * b[idx_r], w_r] x4 -> relu(W_pre g) -> dw = W_dw . + b_dw -> out = sum_r (w_r + dw_r) * scores[idx_r] + bias.
 * b [V, cb] is the per-vertex bottleneck (cb = 8), scores [V, C] the per-vertex class scores,
 * W_pre [4(cb+1), 4(cb+1)], W_dw [4, 4(cb+1)], b_dw [4] torch Linear layouts; d_bias [C] may be NULL. */
int tln_slice_deform(const float* d_b, int cb, const float* d_scores, int64_t V, int C, const int32_t* d_indices,
                     const float* d_weights, const float* d_w_pre, const float* d_w_dw, const float* d_b_dw,
                     const float* d_bias, int64_t n, float* d_out, void* stream);
/* the same, and d_logsm [n, C] = log_softmax(d_out) over the classes (what LNN_SEQ.forward returns beside the raw
 * scores, models.py:466-468); d_logsm may be NULL; C <= 64 */
int tln_slice_deform_ls(const float* d_b, int cb, const float* d_scores, int64_t V, int C, const int32_t* d_indices,
                     const float* d_weights, const float* d_w_pre, const float* d_w_dw, const float* d_b_dw,
                     const float* d_bias, int64_t n, float* d_out, float* d_logsm, void* stream);

/* the heads of n lock-stepped sequences (same head weights) in one launch; d_logsm NULL in all calls or in none */
typedef struct {
  const float* d_b;
  const float* d_scores;
  int64_t V;
  const int32_t* d_indices;
  const float* d_weights;
  int64_t n;
  float* d_out;
  float* d_logsm;
} tln_slice_call;
int tln_slice_deform_multi(const tln_slice_call* calls, int n, int cb, int C, const float* d_w_pre, const float* d_w_dw,
                           const float* d_b_dw, const float* d_bias, void* stream);

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

/* ======================================================================================================
 * Frame program: the per-frame forward of LNN_SEQ (models.py:284-476: distribute, PointNet, U-Net of
 * GroupNorm-ReLU-products, the fusion modules with their hidden states, slice) as TWO native calls per frame
 * instead of ~150 Python-level operator calls.  The host walks its module tree once (after the lazily created
 * parameters exist) and describes the frame as a list of ops over numbered buffers ("slots"); the library sizes
 * the slots from the frame's vertex counts, places them in its own arena and launches the same kernels the
 * operator API launches, in the same order, so both routes give identical results.
 *
 * A slot is [rows, cols] fp32 where rows is symbolic: a lattice level (its current vertex count), the point
 * count N, 4N, or the row count a hidden state had when it was stored (TLN_ROWS_STATE - id).
 * Hidden states (GRUModule.h_lv lm:56/63, CrossframeLocalInterpolationModule.h_lv lm:209/230) live in the
 * program: a STATE_NEW slot is what the module stores this frame, STATE_PREV what it stored on the previous one.
 * An op may be conditional on whether its state exists yet (first frame of a sequence: lm:54, 208).
 * ====================================================================================================== */
typedef struct tln_program tln_program_t;

#define TLN_MAX_LEVELS 8
#define TLN_MAX_STATES 8
#define TLN_ROWS_POINTS (-1)      /* N                                   */
#define TLN_ROWS_POINT_ROWS (-2)  /* 4N                                  */
#define TLN_ROWS_STATE (-16)      /* TLN_ROWS_STATE - id: rows of state id as stored on the previous frame */

enum { TLN_SLOT_F32 = 0, TLN_SLOT_STATS = 1, TLN_SLOT_STATE_NEW = 2, TLN_SLOT_STATE_PREV = 3, TLN_SLOT_OUT = 4 };
typedef struct {
  int rows;   /* >= 0: lattice level; else one of the TLN_ROWS_* codes                              */
  int cols;
  int kind;   /* STATS: [ceil(rows/32)][cols] (sum,sumsq) doubles; OUT: the caller's result buffer   */
  int state;  /* STATE_NEW / STATE_PREV: which hidden state                                          */
} tln_slot;

enum { TLN_TABLE_NONE = 0, TLN_TABLE_NBR = 1, TLN_TABLE_C2F = 2, TLN_TABLE_F2C = 3 };
typedef struct {
  int slot;             /* -1 = absent                                                               */
  int table;            /* TLN_TABLE_*: NONE => row m itself (1 tap), else 9 taps                     */
  int level;            /* NBR: the level itself; C2F / F2C: the COARSE level that owns the table     */
  int relu;             /* ReLU on the operand (after the GroupNorm prologue if any)                  */
  float pad_value;      /* rows past the slot's row count read as this (hidden-state padding)         */
  int gn_stats;         /* slot holding the partial sums of `slot` => GroupNorm prologue, or -1       */
  int gn_groups;
  float gn_eps;
  const float* gn_gamma;
  const float* gn_beta;
} tln_op_src;

enum {
  TLN_OP_GEMM = 1,        /* out[:, out_col:out_col+n] = epi([s0 | s1] @ w); rows = rows(out)            */
  TLN_OP_GN_PARTIALS,     /* stats_out = partial sums of s0.slot                                        */
  TLN_OP_POOL,            /* PointNet pool of the frame's distributed rows: p[0..3] weights, p[4..7] biases,
                             i[0] layers, i[1..5] dims, i[6] min_points                                 */
  TLN_OP_GRU,             /* out = GRUCell(s0.slot, zero-padded s1.slot): p[0..3] = w_ih, w_hh, b_ih, b_hh */
  TLN_OP_AFLOW,           /* out = AFlow(s0.slot, s1.slot padded with f[2]); table NBR of s0.level; f[0] alpha,
                             f[1] beta, i[0] use_center, bias                                            */
  TLN_OP_SLICE_GATHER,    /* out [N, 4*(cols+1)] from s0.slot and the frame's indices / weights          */
  TLN_OP_SLICE,           /* out [N, cols] = blend of s0.slot rows, s1.slot = delta weights (or -1), bias */
  TLN_OP_COPY,            /* out[:, out_col:out_col+cols(s0)] = s0.slot; i[0] != 0: row 0 of the copy zeroed   */
  TLN_OP_ZERO_ROW0,       /* out[0, :] = 0 (lm:569-570)                                                  */
  TLN_OP_STOP_IF_EARLY,   /* early_return frames end here; s0.slot is what the frame returns            */
  TLN_OP_SLICE_DEFORM,    /* out [N, cols(s1)] = tln_slice_deform(b = s0.slot, scores = s1.slot; p[0..2] = W_pre, W_dw,
                             b_dw; bias)                                                                */
  TLN_OP_LSTM_GATES,      /* out [V, C] = tln_lstm_gates(s0.slot [V, 4C])                                */
  TLN_OP_TEMPORAL_MAX,    /* out = max(s1.slot padded with f[0], s0.slot)                                */
  TLN_OP_CGA_GATE,        /* out = sigmoid(s0.slot / (V + C)) * s1.slot, rows past rows(state i[0]) pass unchanged */
  TLN_OP_FILL_EMPTY       /* out = s0.slot with the rows whose first i[0] channels are zero set to f[0]  */
};
typedef struct {
  int kind;
  int cond_state;         /* -1: always; else run only if (state exists) == cond_has                    */
  int cond_has;
  int out, out_col, n;
  int stats_out;          /* GEMM: slot for the partial sums of the output, or -1                        */
  tln_op_src s0, s1;
  int residual;           /* slot or -1                                                                  */
  int relu, w_is_nk;
  const float* w;
  const float* bias;
  const float* p[8];
  int i[8];
  float f[4];
} tln_op;

int tln_program_create(tln_program_t** out, const tln_slot* slots, int n_slots, const tln_op* ops, int n_ops,
                       int n_states, int nr_coarse_levels);
int tln_program_destroy(tln_program_t* p);
/* forget all hidden states (LNN_SEQ.reset_sequence, models.py:252-263) */
int tln_program_reset(tln_program_t* p);
/* K1 of the frame (DistributeLatticeModule models.py:298 + every coarse level and table): v_out receives the
 * vertex count of levels 0..nr_coarse_levels.  The distributed / indices / weights rows stay in the program. */
int tln_program_begin_frame(tln_program_t* p, tln_lattice_t* l, const float* d_positions, const float* d_values,
                            int64_t n, int val_dim, int reset_hashmap, int subtract_mean, int64_t* v_out,
                            void* stream);
/* the same in two halves (tln_distribute_begin / _finish): a lock-step group starts the frames of all its programs
 * before it finishes the first, so that the host waits for vertex counters once instead of once per sequence */
int tln_program_begin_frame_start(tln_program_t* p, tln_lattice_t* l, const float* d_positions, const float* d_values,
                                  int64_t n, int val_dim, int reset_hashmap, int subtract_mean, void* stream);
int tln_program_begin_frame_finish(tln_program_t* p, int64_t* v_out, void* stream);
/* the frames of `count` (1..8) lock-stepped sequences begun together: one batch of K1 launches for all of them
 * (tln_distribute_begin_multi), one wait for the vertex counters; v_out is [count][TLN_MAX_LEVELS].  With stage timing
 * on, programs[0] holds the events around the batched stages (tln_program_timing_read: duration of the batch).
 * need_indices = 0: the frames will return early (tln_program_run_group with early != 0): the per-row vertex indices,
 * which only the slice ops read, are not written. */
int tln_program_begin_frame_group(tln_program_t* const* programs, tln_lattice_t* const* lattices,
                                  const float* const* d_positions, const float* const* d_values, const int64_t* n,
                                  int count, int val_dim, int reset_hashmap, int subtract_mean, int need_indices,
                                  int64_t* v_out, void* stream);
/* the rest of the frame.  early != 0: stop at the program's STOP_IF_EARLY op and copy that slot to d_out (d_out NULL:
 * no copy — the caller drops the early-return value, as the reference's train / test loops do with the tensor
 * models.py:430 hands them); else the TLN_SLOT_OUT slot is d_out.  d_out must hold out_rows x out_cols floats (checked). */
int tln_program_run(tln_program_t* p, int early, float* d_out, int64_t out_rows, int out_cols, void* stream);
/* measurement (bench.py roofline): with capture on, tln_program_run remembers the resolved arguments of every
 * gather-GEMM it launches; replay launches that list `reps` times back to back between two HIP events on `stream`
 * and returns the elapsed milliseconds, the number of launches and their algorithmic flops / bytes.
 * The GRU cell's two projections are on the list too.  The frame's buffers are still in place, so the replays
 * recompute the same values. */
/* one-shot: the slice head of the next tln_program_run also writes log_softmax(scores) to d_logsm [N, classes] */
int tln_program_set_aux_out(tln_program_t* p, float* d_logsm);
int tln_program_capture_gemms(tln_program_t* p, int enable);
/* measurement (bench.py roofline_scatter): with timing on, every frame records HIP events on its launch stream around
 * K1 (all kernels of the distribute), K2 (the PointNet pool) and K8 (the slice kernels); _read waits for them and
 * returns the three durations of the last frame in milliseconds (-1 where the frame had no such stage) */
int tln_program_timing(tln_program_t* p, int enable);
int tln_program_timing_read(tln_program_t* p, float* ms_out /* [3] */);
int tln_program_replay_gemms(tln_program_t* p, int reps, double* ms_total, int64_t* launches, double* flops,
                             double* bytes, void* stream);
/* the same for 1..8 lock-stepped programs (tln_program_run_group): product i of every program goes out through one
 * tln_gather_gemm_multi call, as the group issued it; launches = products */
int tln_program_replay_gemms_group(tln_program_t* const* pp, int n, int reps, double* ms_total, int64_t* launches,
                                   double* flops, double* bytes, void* stream);
/* What the matrix cores EXECUTE for those products (one extra pass of the launches, n = 1..8 programs as above): every
 * gather-GEMM kernel counts its 32 x 32 x 32 multiply steps into a device counter -- gemm_v2 after skipping the K chunks
 * of taps no row of a block has, every kernel including the rows / columns its tiles pad.  flops_executed = steps x 65536;
 * beside the algorithmic 2*M*K*N of tln_program_replay_gemms it says how much of the im2row product's zero work is
 * left. */
int tln_program_replay_executed(tln_program_t* const* pp, int n, double* flops_executed, void* stream);
/* The frame in segments, for the frame-sharded multi-GPU path (temporal_latticenet_amd/dist.py): the rank that owns
 * frame t receives every fusion module's hidden state from the rank of frame t-1 right before the first op that reads
 * it and sends the new one on right after the last op that writes it.  Per frame: tln_program_begin_frame, every state
 * the frame will read announced with its row count (tln_program_state_expect: the sizing walk must see them; the row
 * count of a state is the vertex count of its level before this frame), tln_program_run_begin, then alternately
 * tln_program_run_until(first_read_op) + tln_program_state_set and tln_program_run_until(last_write_op + 1) +
 * tln_program_state_get_new, finally tln_program_run_end.  Same kernels in the same order as tln_program_run. */
int tln_program_nr_ops(const tln_program_t* p);
int tln_program_state_ops(const tln_program_t* p, int id, int* first_read_op, int* last_write_op, int* level);
int tln_program_state_expect(tln_program_t* p, int id, int64_t rows, void* stream);
int tln_program_run_begin(tln_program_t* p, int early, float* d_out, int64_t out_rows, int out_cols, void* stream);
int tln_program_run_until(tln_program_t* p, int op_end, void* stream);
int tln_program_run_end(tln_program_t* p, void* stream);
int tln_program_state_new_info(const tln_program_t* p, int id, int64_t* rows, int* cols, int* written);
int tln_program_state_get_new(tln_program_t* p, int id, float* d_out, void* stream);
/* Pair mode: two programs compiled from the same model (same weights), each with its own open frame
 * (tln_program_begin_frame on its own lattice), stepped in lock-step on ONE stream: every op runs per program, except
 * that the gather-GEMM ops of the two are issued pairwise through tln_gather_gemm_pair.  Results per program as from
 * tln_program_run up to the K-summation order of the paired products.  Both outputs have out_cols columns. */
int tln_program_run_pair(tln_program_t* a, tln_program_t* b, int early, float* d_out_a, int64_t out_rows_a,
                         float* d_out_b, int64_t out_rows_b, int out_cols, void* stream);
/* the same for a group of n = 1..8 programs (tln_gather_gemm_multi) */
int tln_program_run_group(tln_program_t* const* programs, int n, int early, float* const* d_out,
                          const int64_t* out_rows, int out_cols, void* stream);
/* copies *opt (NULL: the defaults) into the program: every product / GRU cell it issues is selected under it, group mode
 * batches what opt->group_off_mask leaves (a group follows its first program's options) */
int tln_program_set_options(tln_program_t* p, const tln_options* opt);
/* device memory the program owns, in bytes: out[0] arena of the frame's temporaries (capacity), [1] its high-water mark in
 * the last frame, [2] the K1 output buffer, [3] the hidden-state buffers, [4] total of 0, 2, 3 */
int tln_program_memory(const tln_program_t* p, int64_t* out /* [5] */);
/* The arena, the hidden-state buffers and every coarse-level temporary of a frame are planned BEFORE the coarse levels'
 * vertex counts have reached the host, with a prediction (old count + max(2048, r_i x new level-0 vertices), r = 1, 1/2,
 * ...) instead of the lattice's hard bound (4^i x new level-0 vertices); a frame whose exact counts exceed the prediction
 * is planned again with them (same results; the arena may grow with its contents kept).  Number of such frames: */
int64_t tln_program_replans(const tln_program_t* p);
/* device pointers of the current frame's K1 outputs ([4N, 3+val_dim+1], [4N], [4N]); valid until the next frame */
int tln_program_frame_rows(tln_program_t* p, const float** d_distributed, const int32_t** d_indices,
                           const float** d_weights, int64_t* rows, int* cols);
/* hidden-state exchange with the operator-level route: rows / existence, copy out, copy in */
int tln_program_state_info(const tln_program_t* p, int id, int64_t* rows, int* cols, int* exists);
int tln_program_state_get(tln_program_t* p, int id, float* d_out, void* stream);
int tln_program_state_set(tln_program_t* p, int id, const float* d_in, int64_t rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif
