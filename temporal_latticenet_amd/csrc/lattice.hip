// Lattice structure for gfx950: hash table with deterministic first-touch numbering,
// K1 distribute, vertex-sorted row list (CSR), neighbour and cross-level tables.
//
// Replaces latticenet.Lattice + DistributeLatticeModule + the structural half of
// GnReluCoarsen/GnReluFinefy (reference call sites: train_ln.py:106,239; models.py:298,353,398;
// lattice_modules.py:285-304).  The arithmetic follows oracle/permuto.py step by step and must
// produce bit-identical integer outputs; this file is therefore compiled with FP contraction OFF.
#pragma clang fp contract(off)
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <math.h>

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void tln_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tln_last_error(void) { return g_err; }
extern "C" int tln_version(void) { return 1; }

// ---------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------
enum { CTR_NV = 0, CTR_NEW = 1, CTR_OVERFLOW = 2, CTR_PROBE_FAIL = 3, CTR_VOLD = 4, CTR_OCCUPIED = 5,
       CTR_CURSOR = 6 /* rows placed in vertex bins */, CTR_TAIL = 7 /* rows without a vertex */,
       CTR_BUCKET_FULL = 8 /* partitioned K1: keys that did not fit a bucket's LDS table (the frame is redone by the
                              per-row-atomic kernels) */,
       CTR_COUNT = 9 };
// Probing stays inside the aligned group of TLN_SLOT_GROUP slots the key hashes into (wraps there).  The table is
// sized for a load of at most 1/2, so a group never fills; what the grouping buys: the buckets of the partitioned K1
// (top bits of the home slot) own disjoint slot ranges, so a bucket's workgroup inserts its keys with plain stores.
#define TLN_SLOT_GROUP 256
#define TLN_MAX_PROBES TLN_SLOT_GROUP
__host__ __device__ __forceinline__ uint64_t tln_next_slot(uint64_t slot) {
  return (slot & ~(uint64_t)(TLN_SLOT_GROUP - 1)) | ((slot + 1) & (uint64_t)(TLN_SLOT_GROUP - 1));
}
#define TLN_SCAN_BLOCK 1024

// One hash-table slot: the packed key, the vertex index (-1 until numbered) and the smallest row id that touched the
// slot while it was un-numbered.  16 bytes, read with ONE load per probe step (three arrays meant three dependent
// cache lines per step).  Empty = all bits set.
struct __attribute__((aligned(16))) TlnSlot {
  unsigned long long key;
  int32_t val;
  uint32_t touch;
};

// partitioned K1 (k_bk_*, below): bucket geometry and the 32-byte row record
#define TLN_BK_HT 1024            // entries of a bucket's LDS hash table (distinct keys of one bucket and frame)
#define TLN_BK_ROWS 512           // rows per bucket aimed at: small enough that a vertex with hundreds of rows (the
                                  // heaviest bucket is ~2000 rows on a KITTI-like frame) does not set the kernel's tail
#define TLN_BK_THREADS 512        // workgroup of a bucket: one run (= one split block's records of the bucket) per thread
#define TLN_BK_MINB 16
#define TLN_BK_MAXB 8192
#define TLN_BK_MAX_PPB 4096   // points of a split block at most
#define TLN_BK_SPLIT_BLOCKS TLN_BK_THREADS   // at most this many split blocks (columns of the bucket-offset table)
// (the row record is a uint4: barycentric weight bits, row id, packed key lo / hi)

struct tln_lattice {
  tln_options opt = [] {   // kernel-selection options of this handle's distributes / pools (tln_lattice_set_options)
    tln_options d;
    tln_options_init(&d);
    return d;
  }();
  int pos_dim = 3, level = 0;
  int64_t capacity = 0, nslots = 0;
  double sigmas[3] = {1, 1, 1};
  float scale[3] = {1, 1, 1};
  double scale_constant = 0;   // c of scale_i = c / (sigma_i sqrt((i+1)(i+2))); tln_lattice_create_ex
  TlnSlot* slots = nullptr;   // [nslots] {key, vertex index, first-touch row}: one 16-byte record per probe step
  int32_t* vkeys = nullptr;  // [capacity][4]
  int32_t* d_ctr = nullptr;
  int32_t* h_ctr = nullptr;
  int32_t* h_ctr_dev = nullptr;   // the same pinned words as the device addresses them
  int64_t nr_vertices = 0, overflow_rows = 0;
  int64_t occupied = 0;  // claimed slots (numbered vertices + keys rejected by the capacity)
  // tables (owned)
  int32_t* nbr = nullptr;
  tln_lattice* coarse = nullptr;
  tln_lattice* parent = nullptr;
  int32_t* c2f = nullptr;
  int32_t* f2c = nullptr;
  // row orders of the three tap tables (rows with equal sets of present taps next to each other: perm section below)
  int32_t *perm_nbr = nullptr, *perm_c2f = nullptr, *perm_f2c = nullptr;
  int64_t table_cap_nbr = 0, table_cap_c2f = 0, table_cap_f2c = 0;   // rows the three tables were allocated for
  int32_t *phist_nbr = nullptr, *phist_c2f = nullptr, *phist_f2c = nullptr;
  int64_t embedded_fine = 0;
  // tln_lattice_prepare_levels_begin without its _finish yet: coarse counters are in flight (root level only)
  int levels_pending = 0;
  hipEvent_t levels_event = nullptr;
  hipEvent_t levels_wait = nullptr;   // a batched _begin: the event of the batch's first lattice (borrowed)
  // tln_distribute_begin .. _finish: the counter fetch in flight and what the second half needs
  hipEvent_t ctr_event = nullptr;
  hipEvent_t ctr_wait = nullptr;   // a batched first half: the event of the batch's first lattice (borrowed)
  bool dist_pending = false;
  int64_t bucket_fallbacks = 0;   // frames redone by the per-row-atomic kernels after a bucket's LDS table overflowed
  const float* dist_pos = nullptr;
  const float* dist_val = nullptr;
  float* dist_w = nullptr;
  float* dist_out = nullptr;
  const int32_t* dist_idx = nullptr;
  int64_t dist_rows = 0;
  int dist_val_dim = 0, dist_subtract = 0;
  // per-call row workspace
  int64_t rows_cap = 0;
  int32_t* row_slot = nullptr;
  int32_t* block_cnt = nullptr;
  int32_t *sk_in = nullptr, *sk_out = nullptr, *sv_in = nullptr, *sv_out = nullptr;
  void* sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  int32_t* seg_start = nullptr;  // [capacity+2]
  float* mean = nullptr;         // [capacity][3] local mean of the last distribute
  long long* pieces = nullptr;   // [rows_cap/256+1][2][3] fixed-point partial sums of segments that span blocks
  int64_t csr_rows = -1;
  // pool workspace
  unsigned long long* pool_packed = nullptr;
  int64_t pool_packed_elems = 0;
  // ---- vertex bins of the last distribute (level 0 only): the frame's rows grouped by vertex WITHOUT a sort.
  // k_distribute_insert counts the rows of every slot (the atomic's return value is the row's rank inside its
  // vertex), k_bins_alloc hands every vertex a contiguous segment, k_bins_scatter moves the row payload there.  The
  // order of the rows inside a segment and of the segments is arbitrary; everything computed from them (max / arg-max
  // with the smallest-row tie rule, fixed-point means) is order-independent.
  uint32_t* slot_cnt = nullptr;   // [nslots] rows of the current frame per slot; zero between frames
  int32_t* vslot = nullptr;       // [capacity] slot of vertex v
  int32_t* row_rank = nullptr;    // [rows_cap]
  int32_t* vcnt = nullptr;        // [capacity] rows of the frame on vertex v
  int32_t* vstart = nullptr;      // [capacity] first bin position of vertex v
  struct TlnBinRec* bin_rec = nullptr;   // [rows_cap] {position, value | barycentric weight, row id, vertex (-1: none), 0}
  int64_t bins_rows = -1;         // rows of the frame the bins hold (-1: none)
  // ---- partitioned K1 (k_bk_*): the rows of a frame split by key hash into buckets, one workgroup per bucket
  float4* posv = nullptr;         // [rows_cap / 4] the frame's points as {x, y, z, value} (partitioned K1)
  uint4* rec = nullptr;           // [rec_cap] 16-byte row records {weight, row, key}, grouped by (split block, bucket)
  int64_t rec_cap = 0;
  uint32_t* bk_off = nullptr;     // [bk_maxb + 1][split blocks] first record of a bucket inside a split block's region
  int bk_maxb = 0;
  uint32_t* first_flag = nullptr; // [rows_cap / 32] bit per row: first-touch row of a key without a vertex (this frame)
  uint32_t* bucket_rows = nullptr;   // [bk_maxb] rows of the frame in a bucket
  uint32_t* bits_pre = nullptr;      // [rows_cap / 128 + 8] set bits of first_flag before every uint4 of it
  int32_t* vstamp = nullptr;      // [capacity] bins_stamp of the last frame that put rows on vertex v
  int bins_stamp = 0;
  bool bins_stamped = false;      // the bins of the last distribute come from the partitioned path (vcnt valid where stamped)
  const float* bins_dist = nullptr;
  const float* bins_weights = nullptr;
  int bins_subtract = 0;
  bool overflow_stale = false;    // CTR_OVERFLOW of the last distribute is still on the device only
  // structure generation: bumped whenever the vertex set may have changed (clear, any insertion that numbered a
  // vertex); the neighbour / cross-level tables remember the generations they were built for
  uint64_t gen = 1;
  uint64_t nbr_gen = 0, c2f_gen_c = 0, c2f_gen_f = 0, f2c_gen_c = 0, f2c_gen_f = 0;
};

// accessors for the other translation units
const int32_t* tln_lat_order(const tln_lattice* l) { return l->sv_out; }
const int32_t* tln_lat_sorted_vertex(const tln_lattice* l) { return l->sk_out; }
const int32_t* tln_lat_seg_start(const tln_lattice* l) { return l->seg_start; }
int64_t tln_lat_csr_rows(const tln_lattice* l) { return l->csr_rows; }
// the pool's packed (value, ~row) accumulators.  INVARIANT: all zero between pool calls (both finalise kernels write the
// zeros back), so no per-frame memset of V x 64 x 8 bytes; sized once for the level's capacity.
int tln_lat_pool_ws(tln_lattice* l, int64_t elems, unsigned long long** out) {
  if (elems > l->pool_packed_elems) {
    if (l->pool_packed) {
      TLN_HIP(hipDeviceSynchronize());
      (void)hipFree(l->pool_packed);
    }
    l->pool_packed = nullptr;
    l->pool_packed_elems = 0;
    // sized for what the level holds (+ half again: the lattice grows from frame to frame), not for its capacity: the
    // capacity is a bound the caller picks generously (cfg:71) and 512 bytes per possible vertex were 128 MB per sequence
    int64_t want = elems + elems / 2;
    if (want < (1 << 20)) want = 1 << 20;
    if (want > l->capacity * 64 && l->capacity * 64 >= elems) want = l->capacity * 64;
    TLN_HIP(hipMalloc(&l->pool_packed, (size_t)want * sizeof(unsigned long long)));
    TLN_HIP(hipMemset(l->pool_packed, 0, (size_t)want * sizeof(unsigned long long)));
    TLN_HIP(hipDeviceSynchronize());
    l->pool_packed_elems = want;
  }
  *out = l->pool_packed;
  return TLN_OK;
}

static void set_vertices(tln_lattice* l, int64_t n) {
  if (n != l->nr_vertices) ++l->gen;
  l->nr_vertices = n;
}

static int bits_for(int64_t v) {
  int b = 1;
  while ((1ll << b) <= v) ++b;
  return b;
}

#define RADIX_KPB 4096   // radix sort: keys per block (csr build below)

static int ensure_rows(tln_lattice* l, int64_t rows) {
  if (rows <= l->rows_cap) return TLN_OK;
  int64_t cap = 1;
  while (cap < rows) cap <<= 1;
  void* ptrs[] = {l->row_slot, l->block_cnt, l->sk_in, l->sk_out, l->sv_in, l->sv_out, l->sort_temp, l->pieces,
                  l->row_rank, l->bin_rec, l->rec, l->bk_off, l->first_flag,
                  l->bucket_rows, l->bits_pre, l->posv};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  l->posv = nullptr;
  l->row_slot = l->block_cnt = l->sk_in = l->sk_out = l->sv_in = l->sv_out = nullptr;
  l->row_rank = nullptr;
  l->bin_rec = nullptr;
  l->rec = nullptr;
  l->bk_off = l->first_flag = l->bucket_rows = l->bits_pre = nullptr;
  l->rec_cap = 0;
  l->bk_maxb = 0;
  l->bins_rows = -1;
  l->sort_temp = nullptr;
  l->pieces = nullptr;
  l->rows_cap = 0;
  TLN_HIP(hipMalloc(&l->row_slot, cap * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->block_cnt, (cap / TLN_SCAN_BLOCK + 2) * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->sk_in, cap * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->sk_out, cap * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->sv_in, cap * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->sv_out, cap * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->pieces, (cap / 256 + 2) * 6 * sizeof(long long)));
  if (l->level == 0) {
    TLN_HIP(hipMalloc(&l->row_rank, cap * sizeof(int32_t)));
    TLN_HIP(hipMalloc(&l->bin_rec, cap * sizeof(TlnBinRec)));
    // partitioned K1: the split blocks' record regions (each rounded up to whole blocks of points), the bucket offsets
    // of every split block, the first-touch flags (zero between frames) and the bucket directories
    l->rec_cap = cap + cap / 128 + 4 * TLN_BK_MAX_PPB;
    l->bk_maxb = (int)(cap / 128 < TLN_BK_MINB ? TLN_BK_MINB : (cap / 128 > TLN_BK_MAXB ? TLN_BK_MAXB : cap / 128));
    TLN_HIP(hipMalloc(&l->rec, (size_t)l->rec_cap * sizeof(uint4)));
    TLN_HIP(hipMalloc(&l->posv, (size_t)(cap / 4 + 1) * sizeof(float4)));
    TLN_HIP(hipMalloc(&l->bk_off, (size_t)TLN_BK_SPLIT_BLOCKS * (l->bk_maxb + 1) * sizeof(uint32_t)));
    TLN_HIP(hipMalloc(&l->first_flag, cap * sizeof(uint32_t)));
    TLN_HIP(hipMemset(l->first_flag, 0, cap * sizeof(uint32_t)));
    TLN_HIP(hipMalloc(&l->bucket_rows, (size_t)l->bk_maxb * sizeof(uint32_t)));
    TLN_HIP(hipMalloc(&l->bits_pre, (size_t)(cap / 128 + 8) * sizeof(uint32_t)));
  }
  // radix-sort scratch: ping-pong keys + values and the [256][blocks] digit histogram
  // + the 256 digit bases and the arrival counter of the fused table scan (counter starts, and is left, at zero)
  const size_t hist_ints = (size_t)256 * (cap / RADIX_KPB + 2);
  const size_t bytes = (size_t)cap * 2 * sizeof(int32_t) + (hist_ints + 256 + 16 + 3 * 256) * sizeof(int32_t);
  TLN_HIP(hipMalloc(&l->sort_temp, bytes));
  TLN_HIP(hipMemset(l->sort_temp, 0, bytes));
  TLN_HIP(hipDeviceSynchronize());
  l->sort_temp_bytes = bytes;
  l->rows_cap = cap;
  return TLN_OK;
}

static int lattice_alloc(tln_lattice** out, int pos_dim, const double* sigmas, int64_t capacity, int level,
                         double scale_constant) {
  TLN_REQUIRE(pos_dim == 3, "pos_dim %d unsupported (only 3)", pos_dim);
  TLN_REQUIRE(capacity >= 16 && capacity <= (1ll << 26), "capacity %lld out of range", (long long)capacity);
  tln_lattice* l = new tln_lattice();
  l->pos_dim = pos_dim;
  l->level = level;
  l->capacity = capacity;
  // level 0 starts with room for one 120k-point frame (4 rows per point) at load factor 1/2; every level
  // grows on demand (ensure_slots), so the slot count never limits which keys are accepted
  const int64_t ns = level == 0 ? (1 << 20) : (1 << 16);
  l->nslots = ns;
  for (int i = 0; i < 3; ++i) {
    l->sigmas[i] = sigmas[i];
    // scale_i = c / (sigma_i sqrt((i+1)(i+2))).  The constant c is a free choice of the un-vendored dependency
    // (README.md:47): default = Adams 2010 §3.1's (d+1)*sqrt(2/3), the choice that meets the reference's sizing hint
    // (seq_config/lnn_train_semantic_kitti.cfg:71, ~10k vertices for a KITTI scan at sigma = 1); a
    // checkpoint trained against a build that drops the factor needs c = 1 (tln_lattice_create_ex)
    l->scale[i] = (float)(scale_constant / (sigmas[i] * sqrt((double)((i + 1) * (i + 2)))));
  }
  l->scale_constant = scale_constant;
  TLN_HIP(hipMalloc(&l->slots, ns * sizeof(TlnSlot)));
  TLN_HIP(hipMalloc(&l->vkeys, capacity * 4 * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->d_ctr, CTR_COUNT * sizeof(int32_t)));
  TLN_HIP(hipHostMalloc(&l->h_ctr, CTR_COUNT * sizeof(int32_t), hipHostMallocMapped));
  TLN_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&l->h_ctr_dev), l->h_ctr, 0));
  TLN_HIP(hipMalloc(&l->seg_start, (capacity + 2) * sizeof(int32_t)));
  TLN_HIP(hipMalloc(&l->mean, capacity * 3 * sizeof(float)));
  if (level == 0) {
    TLN_HIP(hipMalloc(&l->slot_cnt, ns * sizeof(uint32_t)));
    TLN_HIP(hipMalloc(&l->vslot, capacity * sizeof(int32_t)));
    TLN_HIP(hipMalloc(&l->vcnt, capacity * sizeof(int32_t)));
    TLN_HIP(hipMalloc(&l->vstart, capacity * sizeof(int32_t)));
    TLN_HIP(hipMalloc(&l->vstamp, capacity * sizeof(int32_t)));
    TLN_HIP(hipMemset(l->vstamp, 0, capacity * sizeof(int32_t)));
  }
  *out = l;
  return TLN_OK;
}

// every level of a lattice emptied by ONE launch (slot tables to 0xFF.., counters to 0) instead of four memsets each
#define TLN_CLEAR_JOBS 24
struct ClearJobs {
  struct {
    TlnSlot* slots;
    uint32_t* cnt;   // per-slot row counts (level 0), may be NULL
    int32_t* ctr;
    int32_t* host_ctr;   // the host's mapped copy of the counters follows
    int64_t nslots;
  } j[TLN_CLEAR_JOBS];
  int n;
};
__global__ void __launch_bounds__(256) k_clear_levels(ClearJobs jobs) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < jobs.n; ++k) {
    // an empty slot is 16 bytes of ones (key, vertex index -1, touch); two slots per step and thread
    ulonglong2* raw = reinterpret_cast<ulonglong2*>(jobs.j[k].slots);
    uint2* cnt2 = reinterpret_cast<uint2*>(jobs.j[k].cnt);
    const int64_t pairs = jobs.j[k].nslots >> 1;  // slot counts are powers of two
    for (int64_t i = id; i < pairs; i += stride) {
      raw[2 * i] = make_ulonglong2(~0ull, ~0ull);
      raw[2 * i + 1] = make_ulonglong2(~0ull, ~0ull);
      if (cnt2) cnt2[i] = make_uint2(0u, 0u);
    }
    if (id < CTR_COUNT) {
      jobs.j[k].ctr[id] = 0;
      jobs.j[k].host_ctr[id] = 0;
    }
  }
}

// clears the level stacks of n lattices: one launch per TLN_CLEAR_JOBS levels (8 sequences x 3 levels = one launch)
extern "C" int tln_lattice_clear_multi(tln_lattice_t* const* ll, int n, void* stream_) {
  TLN_REQUIRE(ll && n >= 1, "null argument");
  hipStream_t s = (hipStream_t)stream_;
  ClearJobs jobs{};
  for (int i = 0; i < n; ++i) {
    tln_lattice* l = ll[i];
    TLN_REQUIRE(l, "null lattice");
    if (l->levels_pending) {  // a fetch of coarse counters is in flight: let it land before the counters are reset
      if (l->levels_pending > 0 && (l->levels_wait || l->levels_event))
        TLN_HIP(hipEventSynchronize(l->levels_wait ? l->levels_wait : l->levels_event));
      l->levels_wait = nullptr;
      l->levels_pending = 0;
    }
    for (tln_lattice* p = l; p; p = p->coarse) {
      if (jobs.n == TLN_CLEAR_JOBS) {
        hipLaunchKernelGGL(k_clear_levels, dim3(2048), dim3(256), 0, s, jobs);
        jobs.n = 0;
      }
      jobs.j[jobs.n].slots = p->slots;
      jobs.j[jobs.n].cnt = p->slot_cnt;
      jobs.j[jobs.n].ctr = p->d_ctr;
      jobs.j[jobs.n].host_ctr = p->h_ctr_dev;
      jobs.j[jobs.n].nslots = p->nslots;
      ++jobs.n;
      p->nr_vertices = 0;
      p->overflow_rows = 0;
      p->occupied = 0;
      ++p->gen;
      p->bins_rows = -1;
      p->overflow_stale = false;
      p->embedded_fine = 0;
      p->csr_rows = -1;
    }
  }
  if (jobs.n) hipLaunchKernelGGL(k_clear_levels, dim3(2048), dim3(256), 0, s, jobs);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

extern "C" int tln_lattice_clear(tln_lattice_t* l, void* stream_) { return tln_lattice_clear_multi(&l, 1, stream_); }

extern "C" double tln_lattice_default_scale_constant(void) { return 4.0 * sqrt(2.0 / 3.0); }

extern "C" int tln_lattice_create_ex(tln_lattice_t** out, int pos_dim, const double* sigmas, int64_t capacity,
                                     double scale_constant) {
  TLN_REQUIRE(out && sigmas, "null argument");
  if (scale_constant == 0.0) scale_constant = tln_lattice_default_scale_constant();
  TLN_REQUIRE(scale_constant > 0.0 && scale_constant < 1e6, "scale constant %g out of range", scale_constant);
  int rc = lattice_alloc(out, pos_dim, sigmas, capacity, 0, scale_constant);
  if (rc) return rc;
  rc = tln_lattice_clear(*out, nullptr);
  if (rc) return rc;
  TLN_HIP(hipStreamSynchronize(nullptr));
  return TLN_OK;
}

extern "C" int tln_lattice_create(tln_lattice_t** out, int pos_dim, const double* sigmas, int64_t capacity) {
  return tln_lattice_create_ex(out, pos_dim, sigmas, capacity, 0.0);
}

extern "C" double tln_lattice_scale_constant(const tln_lattice_t* l) { return l ? l->scale_constant : 0.0; }

static void perm_forget(const int32_t* table);   // (row orders of the tap tables: below, with the tables)

extern "C" int tln_lattice_destroy(tln_lattice_t* l) {
  if (l && l->ctr_event) {
    (void)hipEventDestroy(l->ctr_event);
    l->ctr_event = nullptr;
  }
  if (l && l->levels_event) {
    (void)hipEventDestroy(l->levels_event);
    l->levels_event = nullptr;
  }
  if (!l) return TLN_OK;
  if (l->coarse) tln_lattice_destroy(l->coarse);
  perm_forget(l->nbr);
  perm_forget(l->c2f);
  perm_forget(l->f2c);
  void* ptrs[] = {l->slots, l->vkeys, l->d_ctr, l->nbr, l->c2f, l->f2c, l->perm_nbr, l->perm_c2f, l->perm_f2c, l->phist_nbr,
                  l->phist_c2f, l->phist_f2c,
                  l->row_slot, l->block_cnt, l->sk_in, l->sk_out, l->sv_in, l->sv_out, l->sort_temp,
                  l->seg_start, l->pool_packed, l->mean, l->pieces, l->slot_cnt, l->vslot, l->vcnt, l->vstart,
                  l->row_rank, l->bin_rec, l->rec, l->bk_off, l->first_flag,
                  l->bucket_rows, l->bits_pre, l->vstamp, l->posv};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (l->h_ctr) (void)hipHostFree(l->h_ctr);
  delete l;
  return TLN_OK;
}

extern "C" int64_t tln_lattice_nr_vertices(const tln_lattice_t* l) { return l ? l->nr_vertices : -1; }
extern "C" int64_t tln_lattice_capacity(const tln_lattice_t* l) { return l ? l->capacity : -1; }
extern "C" int tln_lattice_level(const tln_lattice_t* l) { return l ? l->level : -1; }
extern "C" int64_t tln_lattice_overflow_rows(const tln_lattice_t* lc) {
  if (!lc) return -1;
  tln_lattice_t* l = const_cast<tln_lattice_t*>(lc);
  if (l->overflow_stale) {   // the count of the last distribute was finished on the device after the counter fetch
    int32_t v = 0;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&v, l->d_ctr + CTR_OVERFLOW, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
      return -1;
    l->overflow_rows = v;
    l->overflow_stale = false;
  }
  return l->overflow_rows;
}

// ---------------------------------------------------------------------------------------
// device: table access
// ---------------------------------------------------------------------------------------
struct TableRef {
  TlnSlot* slots;
  int32_t* ctr;
  uint64_t mask;
};
static TableRef table_ref(const tln_lattice* l) {
  return TableRef{l->slots, l->d_ctr, (uint64_t)(l->nslots - 1)};
}

// one probe step = ONE 16-byte load of the slot record
__device__ __forceinline__ void load_slot(const TableRef& t, uint64_t slot, unsigned long long& key, int& val,
                                          uint32_t& touch) {
  const ulonglong2 raw = *reinterpret_cast<const ulonglong2*>(&t.slots[slot]);
  key = raw.x;
  val = (int)(uint32_t)(raw.y & 0xFFFFFFFFull);
  touch = (uint32_t)(raw.y >> 32);
}

// find-or-claim the slot of key K; records the smallest row id touching a not-yet-numbered slot
// (*numbered = the slot already carries a vertex index)
__device__ __forceinline__ int probe_insert(const TableRef& t, uint64_t K, uint32_t id, bool* numbered = nullptr) {
  uint64_t slot = tln_mix64(K) & t.mask;
  for (int probe = 0; probe < TLN_MAX_PROBES; ++probe) {
    unsigned long long cur;
    int val;
    uint32_t touch;
    load_slot(t, slot, cur, val, touch);
    if (cur == TLN_KEY_EMPTY) {
      cur = atomicCAS(&t.slots[slot].key, (unsigned long long)TLN_KEY_EMPTY, (unsigned long long)K);
      if (cur == TLN_KEY_EMPTY) {
        cur = K;
        atomicAdd(&t.ctr[CTR_OCCUPIED], 1);
      }
      val = -1;            // a slot that was empty a moment ago is un-numbered, whoever claimed it
      touch = 0xFFFFFFFFu; // (a stale larger value only costs an atomicMin)
    }
    if (cur == K) {
      if (val < 0 && touch > id) atomicMin(&t.slots[slot].touch, id);
      if (numbered) *numbered = val >= 0;
      return (int)slot;
    }
    slot = tln_next_slot(slot);
  }
  atomicAdd(&t.ctr[CTR_PROBE_FAIL], 1);
  if (numbered) *numbered = false;
  return -1;
}

// read-only lookup -> vertex index or -1
__device__ __forceinline__ int probe_find(const TableRef& t, uint64_t K) {
  uint64_t slot = tln_mix64(K) & t.mask;
  for (int probe = 0; probe < TLN_MAX_PROBES; ++probe) {
    unsigned long long cur;
    int val;
    uint32_t touch;
    load_slot(t, slot, cur, val, touch);
    if (cur == K) return val;
    if (cur == TLN_KEY_EMPTY) return -1;
    slot = tln_next_slot(slot);
  }
  return -1;
}

// ---------------------------------------------------------------------------------------
// device: simplex arithmetic (oracle/permuto.py S2 and S5)
// ---------------------------------------------------------------------------------------
// point -> rem0[4], rank[4], bary[4]   (d = 3)
__device__ __forceinline__ void point_simplex(float x, float y, float z, float s0, float s1, float s2, int rem0[4],
                                              int rank[4], float bary[4]) {
  const float cf0 = x * s0, cf1 = y * s1, cf2 = z * s2;
  float e[4];
  float sm = 0.0f;
  e[3] = sm - 3.0f * cf2;
  sm = sm + cf2;
  e[2] = sm - 2.0f * cf1;
  sm = sm + cf1;
  e[1] = sm - 1.0f * cf0;
  sm = sm + cf0;
  e[0] = sm;
  float rf[4];
  int ssum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = e[i] * 0.25f;
    const float up = ceilf(v) * 4.0f, down = floorf(v) * 4.0f;
    rf[i] = ((up - e[i]) < (e[i] - down)) ? up : down;
    rem0[i] = (int)rf[i];
    ssum += rem0[i];
    rank[i] = 0;
  }
  ssum /= 4;  // exact: the sum is a multiple of 4
  float df[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) df[i] = e[i] - rf[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j) {
      const bool lt = df[i] < df[j];
      rank[i] += lt ? 1 : 0;
      rank[j] += lt ? 0 : 1;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool pf = (ssum > 0) && (rank[i] >= 4 - ssum);
    const bool nf = (ssum < 0) && (rank[i] < -ssum);
    rem0[i] += (nf ? 4 : 0) - (pf ? 4 : 0);
    rank[i] += ssum + (nf ? 4 : 0) - (pf ? 4 : 0);
  }
  float b[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float delta = (e[i] - (float)rem0[i]) * 0.25f;
    const int a = 3 - rank[i], c = 4 - rank[i];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == a) b[j] = b[j] + delta;
      if (j == c) b[j] = b[j] - delta;
    }
  }
  b[0] = b[0] + (1.0f + b[4]);
#pragma unroll
  for (int j = 0; j < 4; ++j) bary[j] = b[j];
}

__device__ __forceinline__ void vertex_key(const int rem0[4], const int rank[4], int r, int& k0, int& k1, int& k2) {
  k0 = rem0[0] + r - ((rank[0] > 3 - r) ? 4 : 0);
  k1 = rem0[1] + r - ((rank[1] > 3 - r) ? 4 : 0);
  k2 = rem0[2] + r - ((rank[2] > 3 - r) ? 4 : 0);
}

// fine key -> the coarse simplex around f/2 (integer arithmetic in units of 1/2)
__device__ __forceinline__ void coarse_simplex(int f0, int f1, int f2, int rem0[4], int rank[4], int bn[4]) {
  int e[4] = {f0, f1, f2, -(f0 + f1 + f2)};
  int r[4];
  int ssum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int down = e[i] & ~7;  // floor to a multiple of 8
    const int up = (e[i] != down) ? down + 8 : down;
    r[i] = ((up - e[i]) < (e[i] - down)) ? up : down;
    ssum += r[i];
    rank[i] = 0;
  }
  ssum /= 8;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j) {
      const bool lt = (e[i] - r[i]) < (e[j] - r[j]);
      rank[i] += lt ? 1 : 0;
      rank[j] += lt ? 0 : 1;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool pf = (ssum > 0) && (rank[i] >= 4 - ssum);
    const bool nf = (ssum < 0) && (rank[i] < -ssum);
    r[i] += (nf ? 8 : 0) - (pf ? 8 : 0);
    rank[i] += ssum + (nf ? 4 : 0) - (pf ? 4 : 0);
  }
  int b[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int delta = e[i] - r[i];
    const int a = 3 - rank[i], c = 4 - rank[i];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == a) b[j] += delta;
      if (j == c) b[j] -= delta;
    }
  }
  b[0] += 8 + b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rem0[i] = r[i] / 2;
    bn[i] = b[i];
  }
}

// ---------------------------------------------------------------------------------------
// slot table growth: probing must never fail, so that ONLY the first-touch numbering decides which
// keys fit into `capacity` (deterministic overflow, identical to the sequential oracle)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rehash(const int32_t* __restrict__ vkeys, int64_t nv, TableRef t,
                                                int32_t* __restrict__ vslot) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const uint64_t K = tln_pack_key(vkeys[4 * v], vkeys[4 * v + 1], vkeys[4 * v + 2]);
  uint64_t slot = tln_mix64(K) & t.mask;
  for (int probe = 0; probe < TLN_MAX_PROBES; ++probe) {
    const uint64_t old = atomicCAS((unsigned long long*)&t.slots[slot].key, (unsigned long long)TLN_KEY_EMPTY,
                                   (unsigned long long)K);
    if (old == TLN_KEY_EMPTY) {
      t.slots[slot].val = (int32_t)v;
      if (vslot) vslot[v] = (int32_t)slot;
      return;
    }
    slot = tln_next_slot(slot);
  }
  atomicAdd(&t.ctr[CTR_PROBE_FAIL], 1);
}

static int ensure_slots(tln_lattice* l, int64_t rows, hipStream_t s) {
  if (2 * (l->occupied + rows) <= l->nslots) return TLN_OK;
  int64_t ns = l->nslots;
  while (ns < 2 * (l->nr_vertices + rows)) ns <<= 1;
  TLN_HIP(hipStreamSynchronize(s));
  TlnSlot* nk = nullptr;
  TLN_HIP(hipMalloc(&nk, ns * sizeof(TlnSlot)));
  TLN_HIP(hipMemsetAsync(nk, 0xFF, ns * sizeof(TlnSlot), s));
  if (l->slot_cnt) {   // between frames every count is zero: the new array just starts that way
    (void)hipFree(l->slot_cnt);
    l->slot_cnt = nullptr;
    TLN_HIP(hipMalloc(&l->slot_cnt, ns * sizeof(uint32_t)));
    TLN_HIP(hipMemsetAsync(l->slot_cnt, 0, ns * sizeof(uint32_t), s));
  }
  (void)hipFree(l->slots);
  l->slots = nk;
  l->nslots = ns;
  if (l->nr_vertices > 0) {
    hipLaunchKernelGGL(k_rehash, dim3((unsigned)tln_cdiv(l->nr_vertices, 256)), dim3(256), 0, s, l->vkeys,
                       l->nr_vertices, table_ref(l), l->vslot);
    TLN_LAUNCH_CHECK();
  }
  // keys that were rejected by the capacity are dropped here; they are retried by later insertions
  const int32_t occ = (int32_t)l->nr_vertices;
  TLN_HIP(hipMemcpyAsync(l->d_ctr + CTR_OCCUPIED, &occ, sizeof(int32_t), hipMemcpyHostToDevice, s));
  TLN_HIP(hipStreamSynchronize(s));
  l->occupied = l->nr_vertices;
  l->h_ctr[CTR_OCCUPIED] = occ;   // (the host's mapped copy follows the device counters)
  ++l->gen;   // slot numbers changed: nothing cached by slot survives (the tables hold vertex indices, rebuilt anyway)
  return TLN_OK;
}

// ---------------------------------------------------------------------------------------
// phase A kernels: compute keys, claim slots
// ---------------------------------------------------------------------------------------
// one thread per ROW (point p = id>>2, simplex vertex r = id&3): the four lanes of a point repeat the cheap simplex
// arithmetic, but every hash probe is its own thread — four times the waves to hide the dependent table reads behind
// (one thread per point: 42 us per 120k-point frame; per row: see DESIGN 6)
__global__ void __launch_bounds__(256) k_distribute_insert(const float* __restrict__ pos, const float* __restrict__ val,
                                                           int64_t n, int val_dim, float s0, float s1, float s2,
                                                           TableRef t, int32_t* __restrict__ row_slot,
                                                           float* __restrict__ weights, float* __restrict__ dist,
                                                           uint32_t* __restrict__ slot_cnt,
                                                           int32_t* __restrict__ row_rank) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live_row = (gid >> 2) < n;
  const int64_t p = live_row ? (gid >> 2) : n - 1;   // lanes past the end shadow the last point and store nothing
  const int r = (int)(gid & 3);
  const float x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
  int rem0[4], rank[4];
  float bary[4];
  point_simplex(x, y, z, s0, s1, s2, rem0, rank, bary);
  const float b = r == 0 ? bary[0] : (r == 1 ? bary[1] : (r == 2 ? bary[2] : bary[3]));
  const int cols = 3 + val_dim + 1;
  int k0, k1, k2;
  vertex_key(rem0, rank, r, k0, k1, k2);
  const uint32_t id = (uint32_t)gid;
  {
    // Rows of neighbouring points mostly share their lattice vertices: only the FIRST lane of every distinct key in
    // the wave probes the table (it also carries the smallest row id of its group, which is what first touch needs),
    // the others take its slot.  On the first frame of a sequence every key is new and the CAS / atomicMin traffic of
    // 480k rows on ~2k slots made this kernel 4x slower than on the later frames.
    const bool valid = live_row && tln_key_in_range(k0, k1, k2);
    const uint64_t K = valid ? tln_pack_key(k0, k1, k2) : 0ull;
    const int lane = threadIdx.x & 63;
    int leader = lane;
    unsigned long long group = 1ull << lane;
    // Which lanes MAY share their key: the lanes that agree on 12 hash bits (12 ballots).  With shuffled points (the
    // training loader, cfg shuffle_points) the 64 rows of a wave almost never collide and the exact grouping below has
    // nothing to do; with points in scan order (valid / test) neighbouring rows share most of their vertices.
    const uint32_t hbits = (uint32_t)(tln_mix64(K) >> 40);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 12; ++bit) {
      const bool one = (hbits >> bit) & 1u;
      const unsigned long long m = __ballot(one);
      peers &= one ? m : ~m;
    }
    unsigned long long todo = __ballot(valid && __popcll(peers) > 1);
    while (todo) {                                        // one round per distinct key among the candidates
      const int first = __builtin_ctzll(todo);
      const uint64_t kf = __shfl(K, first, 64);
      const unsigned long long same = __ballot(valid && K == kf);   // equal keys agree on the hash bits: all in `todo`
      if (valid && K == kf) {
        leader = first;
        group = same;
      }
      todo &= ~same;
    }
    int slot = -1;
    int base = 0;
    if (valid && leader == lane) {
      slot = probe_insert(t, K, id);
      // rows of this frame on the slot: the value the atomic returns numbers the group's rows inside their vertex
      // (the bins of k_bins_scatter; any order will do there)
      if (slot_cnt && slot >= 0) base = (int)atomicAdd(&slot_cnt[slot], (uint32_t)__popcll(group));
    }
    slot = __shfl(slot, leader, 64);
    base = __shfl(base, leader, 64);
    if (live_row) {
      row_slot[id] = valid ? slot : -1;
      weights[id] = b;
      if (row_rank) row_rank[id] = base + __popcll(group & ((1ull << lane) - 1ull));
    }
  }
  if (dist == nullptr) return;   // the caller does not want the [4N, 5] rows (frame program: the pool reads the bins)
  if (val_dim == 1) {
    // the 64 rows of a wave are 64 x 20 contiguous bytes: staged through LDS and written as five fully coalesced
    // 256-byte stores instead of five 20-byte-strided ones
    __shared__ float stage[4][64 * 5];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* st = stage[wid];
    st[lane * 5] = x;
    st[lane * 5 + 1] = y;
    st[lane * 5 + 2] = z;
    st[lane * 5 + 3] = val[p];
    st[lane * 5 + 4] = b;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t wave_row0 = gid - lane;                      // first row of this wave
    const int64_t live = 4 * n - wave_row0 < 64 ? 4 * n - wave_row0 : 64;   // <= 0 for a wave wholly past the end
    float* base = dist + wave_row0 * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int e = k * 64 + lane;
      if (e < live * 5) base[e] = st[e];
    }
    return;
  }
  if (!live_row) return;
  float* row = dist + (int64_t)id * cols;
  row[0] = x;
  row[1] = y;
  row[2] = z;
  for (int c = 0; c < val_dim; ++c) row[3 + c] = val[p * val_dim + c];
  row[3 + val_dim] = b;
}

// coarse embedding of the fine vertices [first, first+count)
// With fine_ctr != NULL `count` is only an upper bound known to the host; the true number of fine vertices is read
// from the fine level's device counters and the rows beyond it are marked empty.
__device__ __forceinline__ void coarsen_insert_body(const int32_t* __restrict__ fine_keys, int64_t first, int64_t count,
                                                    const TableRef& t, int32_t* __restrict__ row_slot,
                                                    const int32_t* __restrict__ fine_ctr, int64_t i) {
  if (i >= count) return;
  if (fine_ctr != nullptr && first + i >= (int64_t)fine_ctr[CTR_NV]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) row_slot[4 * i + r] = -1;
    return;
  }
  const int32_t* fk = fine_keys + 4 * (first + i);
  int rem0[4], rank[4], bn[4];
  coarse_simplex(fk[0], fk[1], fk[2], rem0, rank, bn);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t id = (uint32_t)(4 * i + r);
    int slot = -1;
    if (bn[r] > 0) {
      int k0, k1, k2;
      vertex_key(rem0, rank, r, k0, k1, k2);
      if (tln_key_in_range(k0, k1, k2)) slot = probe_insert(t, tln_pack_key(k0, k1, k2), id);
    }
    row_slot[id] = slot;
  }
}
__global__ void __launch_bounds__(256) k_coarsen_insert(const int32_t* __restrict__ fine_keys, int64_t first,
                                                        int64_t count, TableRef t, int32_t* __restrict__ row_slot,
                                                        const int32_t* __restrict__ fine_ctr) {
  coarsen_insert_body(fine_keys, first, count, t, row_slot, fine_ctr, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// One coarse level of one lattice being extended (embedding of the new fine vertices, then the numbering): the kernels
// below take up to TLN_LVL_MAXJOBS of them per launch (blockIdx.y = lattice) — the lock-stepped sequences of a stream
// extend the same level at the same time, each a chain of two or three launches of a few workgroups otherwise.
#define TLN_LVL_MAXJOBS 8
struct LevelJob {
  const int32_t* fine_keys;
  const int32_t* fine_ctr;
  int32_t* row_slot;
  int32_t* ctr;
  int32_t* host_ctr;     // the host's mapped copy of the counters, written by the numbering (NULL: not wanted)
  int32_t* vkeys;
  int32_t* vslot;
  int32_t* block_cnt;
  TableRef t;
  int64_t first, bound, rows;
  int capacity, vold, nblocks, small;   // small: numbered by ONE workgroup (k_number_small), else count + assign
};
struct LevelJobs {
  LevelJob j[TLN_LVL_MAXJOBS];
};
__global__ void __launch_bounds__(256) k_coarsen_insert_m(const LevelJobs jobs) {
  const LevelJob& J = jobs.j[blockIdx.y];
  coarsen_insert_body(J.fine_keys, J.first, J.bound, J.t, J.row_slot, J.fine_ctr, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void __launch_bounds__(256) k_insert_keys(const int32_t* __restrict__ keys, int64_t n, TableRef t,
                                                     int32_t* __restrict__ row_slot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k0 = keys[3 * i], k1 = keys[3 * i + 1], k2 = keys[3 * i + 2];
  int slot = -1;
  if (tln_key_in_range(k0, k1, k2)) slot = probe_insert(t, tln_pack_key(k0, k1, k2), (uint32_t)i);
  row_slot[i] = slot;
}

// ---------------------------------------------------------------------------------------
// phase B: number the new slots in first-touch order (ballot + prefix sums)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_first_touch(const TableRef& t, const int32_t* row_slot, int64_t id, int64_t rows,
                                               int& slot) {
  slot = -1;
  if (id >= rows) return false;
  slot = row_slot[id];
  if (slot < 0) return false;
  return t.slots[slot].val < 0 && t.slots[slot].touch == (uint32_t)id;
}

__device__ __forceinline__ void publish_counters(const int32_t* ctr, int32_t* host_ctr) {
  if (host_ctr == nullptr) return;
  __threadfence();
  for (int i = 0; i < CTR_COUNT; ++i) host_ctr[i] = __hip_atomic_load(ctr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void count_new_body(const TableRef& t, const int32_t* __restrict__ row_slot, int64_t rows,
                                               int32_t* __restrict__ block_cnt, int bx) {
  __shared__ int wave_cnt[TLN_SCAN_BLOCK / 64];
  if (bx == 0 && threadIdx.x == 0) {   // the bin allocator of this frame starts from zero
    t.ctr[CTR_CURSOR] = 0;
    t.ctr[CTR_TAIL] = 0;
  }
  const int64_t id = (int64_t)bx * TLN_SCAN_BLOCK + threadIdx.x;
  int slot;
  const bool f = is_first_touch(t, row_slot, id, rows, slot);
  const unsigned long long m = __ballot(f);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < TLN_SCAN_BLOCK / 64; ++w) s += wave_cnt[w];
    block_cnt[bx] = s;
  }
}
__global__ void __launch_bounds__(TLN_SCAN_BLOCK) k_count_new(TableRef t, const int32_t* __restrict__ row_slot,
                                                              int64_t rows, int32_t* __restrict__ block_cnt) {
  count_new_body(t, row_slot, rows, block_cnt, (int)blockIdx.x);
}
__global__ void __launch_bounds__(TLN_SCAN_BLOCK) k_count_new_m(const LevelJobs jobs) {
  const LevelJob& J = jobs.j[blockIdx.y];
  if (J.small || (int)blockIdx.x >= J.nblocks) return;
  count_new_body(J.t, J.row_slot, J.rows, J.block_cnt, (int)blockIdx.x);
}

// numbering of the new slots: rank of a first-touch row = first-touch rows before it.  Every block sums the counts of
// the blocks before it itself (a few hundred integers) — no separate scan launch; the last block also publishes the
// new vertex count.  `vold` is the vertex count before this insertion (exact on the host).
__device__ __forceinline__ void assign_new_body(const TableRef& t, const int32_t* __restrict__ row_slot, int64_t rows,
                                                const int32_t* __restrict__ block_cnt, int32_t* __restrict__ ctr, int vold,
                                                int capacity, int32_t* __restrict__ vkeys, int32_t* __restrict__ vslot,
                                                int bx, int nblocks, int32_t* host_ctr) {
  __shared__ int wave_cnt[TLN_SCAN_BLOCK / 64];
  __shared__ int wave_pre[TLN_SCAN_BLOCK / 64];
  const int64_t id = (int64_t)bx * TLN_SCAN_BLOCK + threadIdx.x;
  int slot;
  const bool f = is_first_touch(t, row_slot, id, rows, slot);
  const unsigned long long m = __ballot(f);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int pre = 0;
  for (int bq = threadIdx.x; bq < bx; bq += TLN_SCAN_BLOCK) pre += block_cnt[bq];
  pre = tln_wave_sum(pre);
  if (lane == 0) {
    wave_cnt[wid] = __popcll(m);
    wave_pre[wid] = pre;
  }
  __syncthreads();
  int block_off = 0, block_total = 0;
  for (int w = 0; w < TLN_SCAN_BLOCK / 64; ++w) {
    block_off += wave_pre[w];
    block_total += wave_cnt[w];
  }
  if (bx == nblocks - 1 && threadIdx.x == 0) {
    const int total = block_off + block_total;
    ctr[CTR_VOLD] = vold;
    ctr[CTR_NEW] = total;
    ctr[CTR_OVERFLOW] = 0;  // accumulated by k_row_indices, which runs after the numbering
    const long long vnew = (long long)vold + total;
    ctr[CTR_NV] = (int)(vnew < capacity ? vnew : capacity);
    publish_counters(ctr, host_ctr);
  }
  if (!f) return;
  int woff = 0;
  for (int w = 0; w < wid; ++w) woff += wave_cnt[w];
  const int rank = block_off + woff + __popcll(m & ((1ull << lane) - 1ull));
  const long long v = (long long)vold + rank;
  if (v < capacity) {
    t.slots[slot].val = (int)v;
    if (vslot) vslot[v] = slot;
    int k0, k1, k2;
    tln_unpack_key(t.slots[slot].key, k0, k1, k2);
    int4 kk = make_int4(k0, k1, k2, -(k0 + k1 + k2));
    *reinterpret_cast<int4*>(vkeys + 4 * v) = kk;
  } else {
    t.slots[slot].touch = 0xFFFFFFFFu;  // stays un-numbered; may be retried by a later insertion
  }
}
__global__ void __launch_bounds__(TLN_SCAN_BLOCK) k_assign_new(TableRef t, const int32_t* __restrict__ row_slot,
                                                               int64_t rows, const int32_t* __restrict__ block_cnt,
                                                               int32_t* __restrict__ ctr, int vold, int capacity,
                                                               int32_t* __restrict__ vkeys,
                                                               int32_t* __restrict__ vslot) {
  assign_new_body(t, row_slot, rows, block_cnt, ctr, vold, capacity, vkeys, vslot, (int)blockIdx.x, (int)gridDim.x, nullptr);
}
__global__ void __launch_bounds__(TLN_SCAN_BLOCK) k_assign_new_m(const LevelJobs jobs) {
  const LevelJob& J = jobs.j[blockIdx.y];
  if (J.small || (int)blockIdx.x >= J.nblocks) return;
  assign_new_body(J.t, J.row_slot, J.rows, J.block_cnt, J.ctr, J.vold, J.capacity, J.vkeys, J.vslot, (int)blockIdx.x,
                  J.nblocks, J.host_ctr);
}

// phase C: per-row vertex index (+ sort input for the CSR)
__global__ void __launch_bounds__(256) k_row_indices(TableRef t, const int32_t* __restrict__ row_slot, int64_t rows,
                                                     int32_t* __restrict__ indices, int32_t* __restrict__ ctr,
                                                     int32_t* __restrict__ sk_in, int32_t* __restrict__ sv_in,
                                                     int32_t* __restrict__ sort_totals) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sort_totals)                                        // digit totals of the radix passes that follow
    for (int64_t i = id; i < 3 * 256; i += (int64_t)gridDim.x * blockDim.x) sort_totals[i] = 0;
  if (id >= rows) return;
  const int slot = row_slot[id];
  int idx = -1;
  if (slot >= 0) idx = t.slots[slot].val;
  if (indices) indices[id] = idx;
  if (idx < 0) atomicAdd(&ctr[CTR_OVERFLOW], 1);
  if (sk_in) {
    sk_in[id] = idx < 0 ? ctr[CTR_NV] : idx;  // tail bucket V for rejected rows
    sv_in[id] = (int32_t)id;
  }
}

// small insertions (the coarse levels embed only the new fine vertices of a frame): count, scan and assign in ONE
// launch of a single block that walks the rows in order
__device__ __forceinline__ void number_small_body(const TableRef& t, const int32_t* __restrict__ row_slot, int64_t rows,
                                                  int32_t* __restrict__ ctr, int capacity, int32_t* __restrict__ vkeys,
                                                  int32_t* __restrict__ vslot, int32_t* host_ctr) {
  __shared__ int wave_cnt[TLN_SCAN_BLOCK / 64];
  __shared__ int running_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int vold = ctr[CTR_NV];
  if (threadIdx.x == 0) {
    running_s = 0;
    ctr[CTR_CURSOR] = 0;
    ctr[CTR_TAIL] = 0;
  }
  __syncthreads();
  for (int64_t base = 0; base < rows; base += TLN_SCAN_BLOCK) {
    const int64_t id = base + threadIdx.x;
    int slot;
    const bool f = is_first_touch(t, row_slot, id, rows, slot);
    const unsigned long long m = __ballot(f);
    if (lane == 0) wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < TLN_SCAN_BLOCK / 64; ++w) {
      if (w < wid) woff += wave_cnt[w];
      total += wave_cnt[w];
    }
    const int running = running_s;
    if (f) {
      const long long v = (long long)vold + running + woff + __popcll(m & ((1ull << lane) - 1ull));
      if (v < capacity) {
        t.slots[slot].val = (int)v;
        if (vslot) vslot[v] = slot;
        int k0, k1, k2;
        tln_unpack_key(t.slots[slot].key, k0, k1, k2);
        *reinterpret_cast<int4*>(vkeys + 4 * v) = make_int4(k0, k1, k2, -(k0 + k1 + k2));
      } else {
        t.slots[slot].touch = 0xFFFFFFFFu;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) running_s = running + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int total = running_s;
    ctr[CTR_VOLD] = vold;
    ctr[CTR_NEW] = total;
    ctr[CTR_OVERFLOW] = 0;  // accumulated by k_row_indices, which runs after the numbering
    const long long vnew = (long long)vold + total;
    ctr[CTR_NV] = (int)(vnew < capacity ? vnew : capacity);
    publish_counters(ctr, host_ctr);
  }
}
__global__ void __launch_bounds__(TLN_SCAN_BLOCK) k_number_small(TableRef t, const int32_t* __restrict__ row_slot,
                                                                 int64_t rows, int32_t* __restrict__ ctr, int capacity,
                                                                 int32_t* __restrict__ vkeys,
                                                                 int32_t* __restrict__ vslot) {
  number_small_body(t, row_slot, rows, ctr, capacity, vkeys, vslot, nullptr);
}
__global__ void __launch_bounds__(TLN_SCAN_BLOCK) k_number_small_m(const LevelJobs jobs) {
  const LevelJob& J = jobs.j[blockIdx.y];
  if (!J.small || J.rows <= 0) return;
  number_small_body(J.t, J.row_slot, J.rows, J.ctr, J.capacity, J.vkeys, J.vslot, J.host_ctr);
}

static int number_new(tln_lattice* l, int64_t rows, hipStream_t s) {
  const int nblocks = (int)tln_cdiv(rows, TLN_SCAN_BLOCK);
  TableRef t = table_ref(l);
  if (rows <= 16 * TLN_SCAN_BLOCK) {
    hipLaunchKernelGGL(k_number_small, dim3(1), dim3(TLN_SCAN_BLOCK), 0, s, t, l->row_slot, rows, l->d_ctr,
                       (int)l->capacity, l->vkeys, l->vslot);
    TLN_LAUNCH_CHECK();
    return TLN_OK;
  }
  hipLaunchKernelGGL(k_count_new, dim3(nblocks), dim3(TLN_SCAN_BLOCK), 0, s, t, l->row_slot, rows, l->block_cnt);
  hipLaunchKernelGGL(k_assign_new, dim3(nblocks), dim3(TLN_SCAN_BLOCK), 0, s, t, l->row_slot, rows, l->block_cnt,
                     l->d_ctr, (int)l->nr_vertices, (int)l->capacity, l->vkeys, l->vslot);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

static int fetch_counters(tln_lattice* l, hipStream_t s) {
  TLN_HIP(hipMemcpyAsync(l->h_ctr, l->d_ctr, CTR_COUNT * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  TLN_HIP(hipStreamSynchronize(s));
  set_vertices(l, l->h_ctr[CTR_NV]);
  l->occupied = l->h_ctr[CTR_OCCUPIED];
  if (l->h_ctr[CTR_PROBE_FAIL] != 0) {
    tln_set_error("hash probing failed for %d rows (table too full)", l->h_ctr[CTR_PROBE_FAIL]);
    return TLN_E_CAPACITY;
  }
  return TLN_OK;
}

// ---------------------------------------------------------------------------------------
// CSR: rows sorted by vertex (stable), segment starts by binary search
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_seg_start(const int32_t* __restrict__ sorted_keys, int64_t rows, int64_t nv,
                                                   int32_t* __restrict__ seg_start) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > nv + 1) return;
  if (v == nv + 1) {
    seg_start[v] = (int32_t)rows;
    return;
  }
  int64_t lo = 0, hi = rows;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < (int32_t)v) lo = mid + 1;
    else hi = mid;
  }
  seg_start[v] = (int32_t)lo;
}

// ---------------------------------------------------------------------------------------
// stable LSD radix sort of (vertex index, row id) pairs, 8 bits per pass, two launches per pass:
//   k_radix_hist    per-block digit counts -> hist[digit][block]; the last block to arrive scans the table
//   k_radix_scatter ranks inside a wave by ballot multi-split (8 ballots), waves ordered through LDS,
//                   blocks ordered through the scanned table => stable
// The keys are vertex indices (< V+1), so a 120k-point frame on a few thousand vertices needs two passes.
// ---------------------------------------------------------------------------------------
#define RADIX_TPB 1024   // threads per block (16 waves)

// Exclusive scan of the digit-major table hist[256][nblk] by one 1024-thread block: wave w owns digit rows
// 16w..16w+15, scans each along the blocks (coalesced, 64 entries per step) and leaves the within-row exclusive
// scan in place; the 256 row totals are then scanned into dbase[256]. The scatter adds the two.
__device__ __forceinline__ void radix_scan_table(int32_t* __restrict__ hist, int nblk, int32_t* __restrict__ dbase,
                                                 int* tot /* LDS [256] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (nblk <= 128) {
    // the usual case (<= 524k rows): all 32 loads of the wave's 16 rows in flight at once
    int v0[16], v1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int32_t* row = hist + (int64_t)(wid * 16 + r) * nblk;
      v0[r] = (lane < nblk) ? __hip_atomic_load(row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      v1[r] = (lane + 64 < nblk) ? __hip_atomic_load(row + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int32_t* row = hist + (int64_t)(wid * 16 + r) * nblk;
      int i0 = v0[r], i1 = v1[r];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u0 = __shfl_up(i0, o, 64), u1 = __shfl_up(i1, o, 64);
        if (lane >= o) {
          i0 += u0;
          i1 += u1;
        }
      }
      const int c0 = __shfl(i0, 63, 64);
      if (lane < nblk) row[lane] = i0 - v0[r];
      if (lane + 64 < nblk) row[lane + 64] = c0 + i1 - v1[r];
      if (lane == 63) tot[wid * 16 + r] = c0 + i1;
    }
  } else {
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
      const int d = wid * 16 + r;
      int32_t* row = hist + (int64_t)d * nblk;
      int carry = 0;
      for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const int v = (b < nblk) ? __hip_atomic_load(row + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int u = __shfl_up(incl, o, 64);
          if (lane >= o) incl += u;
        }
        if (b < nblk) row[b] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
      }
      if (lane == 0) tot[d] = carry;
    }
  }
  __syncthreads();
  if (wid == 0) {
    // 256 totals, 4 per lane, exclusive
    const int t0 = tot[4 * lane], t1 = tot[4 * lane + 1], t2 = tot[4 * lane + 2], t3 = tot[4 * lane + 3];
    const int s4 = t0 + t1 + t2 + t3;
    int incl = s4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    const int e = incl - s4;
    dbase[4 * lane] = e;
    dbase[4 * lane + 1] = e + t0;
    dbase[4 * lane + 2] = e + t0 + t1;
    dbase[4 * lane + 3] = e + t0 + t1 + t2;
  }
}

// per-block digit counts -> hist[digit][block]; the block that arrives last scans the table (no extra launch)
__global__ void __launch_bounds__(RADIX_TPB) k_radix_hist(const int32_t* __restrict__ keys, int64_t n, int shift,
                                                          int nblk, int32_t* __restrict__ hist,
                                                          int32_t* __restrict__ dbase, unsigned* __restrict__ arrive) {
  __shared__ int cnt[256];
  __shared__ int is_last;
  if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RADIX_KPB;
#pragma unroll
  for (int it = 0; it < RADIX_KPB / RADIX_TPB; ++it) {
    const int64_t i = base + it * RADIX_TPB + threadIdx.x;
    if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255], 1);
  }
  __syncthreads();
  if (threadIdx.x < 256)
    __hip_atomic_store(&hist[threadIdx.x * nblk + blockIdx.x], cnt[threadIdx.x], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's counts are in L2 before thread 0 announces
  __syncthreads();
  // the counts above are agent-scope atomic stores (written through to the point of coherence) and the scan reads
  // them with agent-scope atomic loads, so the hand-off needs no cache-wide release / acquire (each costs several
  // microseconds on the 8-XCD part): completed stores -> barrier -> relaxed arrival count is enough
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (prev == (unsigned)nblk - 1u);
  }
  __syncthreads();
  if (!is_last) return;
  radix_scan_table(hist, nblk, dbase, cnt);
  if (threadIdx.x == 0) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// few blocks (<= 256, i.e. up to a million rows): NO table scan at all.  The histogram kernel writes its counts
// block-major and adds them to 256 global digit totals (integer atomics: order-independent); every scatter block
// then derives its own bases: digit prefix from the totals + the counts of the blocks before it.
__global__ void __launch_bounds__(RADIX_TPB) k_radix_hist_t(const int32_t* __restrict__ keys, int64_t n, int shift,
                                                            int32_t* __restrict__ hist_t, int32_t* __restrict__ total) {
  __shared__ int cnt[256];
  if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RADIX_KPB;
#pragma unroll
  for (int it = 0; it < RADIX_KPB / RADIX_TPB; ++it) {
    const int64_t i = base + it * RADIX_TPB + threadIdx.x;
    if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255], 1);
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int c = cnt[threadIdx.x];
    hist_t[blockIdx.x * 256 + threadIdx.x] = c;
    if (c) atomicAdd(&total[threadIdx.x], c);
  }
}

template <bool BLOCK_MAJOR>
__global__ void __launch_bounds__(RADIX_TPB) k_radix_scatter(const int32_t* __restrict__ keys_in,
                                                             const int32_t* __restrict__ vals_in, int64_t n, int shift,
                                                             int nblk, const int32_t* __restrict__ hist,
                                                             const int32_t* __restrict__ dbase,
                                                             int32_t* __restrict__ keys_out,
                                                             int32_t* __restrict__ vals_out) {
  __shared__ int base[256];            // running output position per digit for this block
  __shared__ int wcount[16][256];      // per-wave digit counts, then per-wave bases
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (!BLOCK_MAJOR) {
    if (threadIdx.x < 256) base[threadIdx.x] = hist[threadIdx.x * nblk + blockIdx.x] + dbase[threadIdx.x];
  } else {
    // hist = [block][256] counts, dbase = the 256 digit totals.  Thread (q, d) sums the blocks b' = q, q+4, ... before
    // this one; the four quarters and the exclusive digit prefix are combined through LDS (wcount is free here).
    int* part = &wcount[0][0];         // [4][256] quarter sums, then [256] digit totals at +1024
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    int run = 0;
    for (int b = q; b < (int)blockIdx.x; b += 4) run += hist[b * 256 + d];
    part[q * 256 + d] = run;
    if (threadIdx.x < 256) part[1024 + d] = dbase[d];
    __syncthreads();
    // exclusive prefix of the digit totals: threads 0..255 = 4 waves of 64 digits
    int t = 0, incl = 0;
    if (threadIdx.x < 256) {
      t = part[1024 + d];
      incl = t;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
      }
      if (lane == 63) part[1280 + wid] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      int woff = 0;
      for (int w = 0; w < wid; ++w) woff += part[1280 + w];
      base[d] = woff + incl - t + (part[d] + part[256 + d]) + (part[512 + d] + part[768 + d]);
    }
    __syncthreads();                   // wcount is reused below
  }
  const int64_t blk0 = (int64_t)blockIdx.x * RADIX_KPB;
  for (int it = 0; it < RADIX_KPB / RADIX_TPB; ++it) {
    for (int k = threadIdx.x; k < 16 * 256; k += RADIX_TPB) (&wcount[0][0])[k] = 0;
    __syncthreads();
    const int64_t i = blk0 + it * RADIX_TPB + threadIdx.x;
    const bool ok = i < n;
    const int key = ok ? keys_in[i] : 0;
    const int val = ok ? vals_in[i] : 0;
    const int d = (key >> shift) & 255;
    // lanes of this wave holding the same digit (invalid lanes form their own class)
    unsigned long long peers = __ballot(ok);
    if (!ok) peers = ~peers;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (ok && rank == 0) wcount[wid][d] = __popcll(peers);
    __syncthreads();
    if (threadIdx.x < 256) {
      int run = base[threadIdx.x];
      for (int w = 0; w < 16; ++w) {
        const int t = wcount[w][threadIdx.x];
        wcount[w][threadIdx.x] = run;
        run += t;
      }
      base[threadIdx.x] = run;
    }
    __syncthreads();
    if (ok) {
      const int pos = wcount[wid][d] + rank;
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __syncthreads();
  }
}

// scratch behind the ping-pong buffers: hist | dbase[256] | arrive[16] | totals[3][256]
static int32_t* sort_hist(tln_lattice* l) { return reinterpret_cast<int32_t*>(l->sort_temp) + 2 * l->rows_cap; }
static int32_t* sort_totals(tln_lattice* l) { return sort_hist(l) + (size_t)256 * (l->rows_cap / RADIX_KPB + 2) + 256 + 16; }

static int radix_sort_pairs(tln_lattice* l, int64_t rows, int bits, hipStream_t s) {
  const int nblk = (int)tln_cdiv(rows, RADIX_KPB);
  int32_t* tmp_k = reinterpret_cast<int32_t*>(l->sort_temp);
  int32_t* tmp_v = tmp_k + l->rows_cap;
  int32_t* hist = tmp_v + l->rows_cap;
  int32_t* dbase = hist + (size_t)256 * (l->rows_cap / RADIX_KPB + 2);
  unsigned* arrive = reinterpret_cast<unsigned*>(dbase + 256);
  int32_t* totals = sort_totals(l);
  const int passes = (bits + 7) / 8;
  // ping-pong so that the LAST pass lands in sk_out / sv_out
  const int32_t* src_k = l->sk_in;
  const int32_t* src_v = l->sv_in;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) % 2) == 0;
    int32_t* dst_k = to_out ? l->sk_out : tmp_k;
    int32_t* dst_v = to_out ? l->sv_out : tmp_v;
    if (nblk <= 256 && passes <= 3) {
      // totals[p] were zeroed by the kernel that produced the sort input (k_row_indices / k_csr_input)
      int32_t* total = totals + 256 * p;
      hipLaunchKernelGGL(k_radix_hist_t, dim3(nblk), dim3(RADIX_TPB), 0, s, src_k, rows, 8 * p, hist, total);
      hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nblk), dim3(RADIX_TPB), 0, s, src_k, src_v, rows, 8 * p, nblk, hist,
                         total, dst_k, dst_v);
    } else {
      hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(RADIX_TPB), 0, s, src_k, rows, 8 * p, nblk, hist, dbase, arrive);
      hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nblk), dim3(RADIX_TPB), 0, s, src_k, src_v, rows, 8 * p, nblk, hist,
                         dbase, dst_k, dst_v);
    }
    src_k = dst_k;
    src_v = dst_v;
  }
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

static int build_csr_sorted(tln_lattice* l, int64_t rows, hipStream_t s) {
  const int bits = bits_for(l->nr_vertices + 1);
  int rc = radix_sort_pairs(l, rows, bits, s);
  if (rc) return rc;
  const int64_t nv = l->nr_vertices;
  hipLaunchKernelGGL(k_seg_start, dim3((unsigned)tln_cdiv(nv + 2, 256)), dim3(256), 0, s, l->sk_out, rows, nv,
                     l->seg_start);
  TLN_LAUNCH_CHECK();
  l->csr_rows = rows;
  return TLN_OK;
}

__global__ void __launch_bounds__(256) k_csr_input(const int32_t* __restrict__ indices, int64_t rows, int32_t nv,
                                                   int32_t* __restrict__ sk_in, int32_t* __restrict__ sv_in,
                                                   int32_t* __restrict__ sort_totals) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = id; i < 3 * 256; i += (int64_t)gridDim.x * blockDim.x) sort_totals[i] = 0;  // radix digit totals
  if (id >= rows) return;
  const int idx = indices[id];
  sk_in[id] = (idx < 0 || idx >= nv) ? nv : idx;
  sv_in[id] = (int32_t)id;
}

extern "C" int tln_build_csr(tln_lattice_t* l, const int32_t* d_indices, int64_t rows, void* stream_) {
  TLN_REQUIRE(l && d_indices && rows > 0 && rows < (1ll << 31), "bad csr arguments");
  hipStream_t s = (hipStream_t)stream_;
  int rc = ensure_rows(l, rows);
  if (rc) return rc;
  hipLaunchKernelGGL(k_csr_input, dim3((unsigned)tln_cdiv(rows, 256)), dim3(256), 0, s, d_indices, rows,
                     (int32_t)l->nr_vertices, l->sk_in, l->sv_in, sort_totals(l));
  TLN_LAUNCH_CHECK();
  return build_csr_sorted(l, rows, s);
}

// read the CSR back (tests, debugging): order[rows], sorted_vertex[rows] (tail bucket = V), seg_start[V+2]
extern "C" int tln_lattice_csr(tln_lattice_t* l, int32_t* d_order, int32_t* d_sorted_vertex, int32_t* d_seg_start,
                               int64_t* rows_out, void* stream_) {
  TLN_REQUIRE(l && rows_out, "bad csr read-out arguments");
  hipStream_t s = (hipStream_t)stream_;
  *rows_out = l->csr_rows;
  if (l->csr_rows <= 0) return TLN_OK;
  if (d_order) TLN_HIP(hipMemcpyAsync(d_order, l->sv_out, l->csr_rows * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (d_sorted_vertex)
    TLN_HIP(hipMemcpyAsync(d_sorted_vertex, l->sk_out, l->csr_rows * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (d_seg_start)
    TLN_HIP(hipMemcpyAsync(d_seg_start, l->seg_start, (l->nr_vertices + 2) * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, s));
  return TLN_OK;
}

// ---------------------------------------------------------------------------------------
// K1 phase D: local mean per vertex and subtraction, independent of how skewed the rows-per-vertex
// distribution is (one vertex next to the sensor collects thousands of rows):
//   k_mean_pieces : 256 consecutive SORTED rows per block, segmented inclusive scan by vertex of the positions in FIXED POINT (int64, units of
//                   2^-20: exact, hence independent of the summation order and bit-identical to oracle/ops.py:distribute);
//                   a segment that lies inside one block is finished there, otherwise the block stores the sum of
//                   its first piece (slot 0, touches the block start) and last piece (slot 1)
//   k_mean_combine: segments spanning several blocks add their pieces in block order
//   k_subtract_rows: dist[row][0:3] = pos - mean[vertex]
// ---------------------------------------------------------------------------------------
#define MEAN_BLOCK 256
__global__ void __launch_bounds__(MEAN_BLOCK) k_mean_pieces(const float* __restrict__ pos,
                                                            const int32_t* __restrict__ order,
                                                            const int32_t* __restrict__ sorted_vertex,
                                                            const int32_t* __restrict__ seg_start, int nv,
                                                            float* __restrict__ mean, long long* __restrict__ pieces) {
  __shared__ long long sx[MEAN_BLOCK], sy[MEAN_BLOCK], sz[MEAN_BLOCK];
  __shared__ int key[MEAN_BLOCK];
  const int valid_rows = seg_start[nv];  // rows that have a vertex (the tail bucket is excluded)
  const int j = threadIdx.x;
  const int base = blockIdx.x * MEAN_BLOCK;
  const int gi = base + j;
  const bool ok = gi < valid_rows;
  int v = -1 - j;  // unique dummy keys for the padding lanes
  long long x = 0, y = 0, z = 0;
  if (ok) {
    v = sorted_vertex[gi];
    const int64_t p = order[gi] >> 2;
    x = tln_fix20(pos[3 * p]);
    y = tln_fix20(pos[3 * p + 1]);
    z = tln_fix20(pos[3 * p + 2]);
  }
  key[j] = v;
  sx[j] = x;
  sy[j] = y;
  sz[j] = z;
  __syncthreads();
#pragma unroll
  for (int o = 1; o < MEAN_BLOCK; o <<= 1) {
    long long ax = 0, ay = 0, az = 0;
    const bool take = (j >= o) && (key[j - o] == v);
    if (take) {
      ax = sx[j - o];
      ay = sy[j - o];
      az = sz[j - o];
    }
    __syncthreads();
    if (take) {
      sx[j] += ax;
      sy[j] += ay;
      sz[j] += az;
    }
    __syncthreads();
  }
  if (!ok) return;
  const int last = ((valid_rows - base) < MEAN_BLOCK ? (valid_rows - base) : MEAN_BLOCK) - 1;
  const bool seg_end = (j == last) || (key[j + 1] != v);
  if (!seg_end) return;
  const int b = seg_start[v], e = seg_start[v + 1];
  if (b >= base && e <= base + last + 1) {
    const double cnt = (double)(e - b);
    mean[3 * v] = tln_unfix20(sx[j], cnt);
    mean[3 * v + 1] = tln_unfix20(sy[j], cnt);
    mean[3 * v + 2] = tln_unfix20(sz[j], cnt);
  } else {
    const int slot = (key[0] == v) ? 0 : 1;
    long long* d = pieces + ((int64_t)blockIdx.x * 2 + slot) * 3;
    d[0] = sx[j];
    d[1] = sy[j];
    d[2] = sz[j];
  }
}

__global__ void __launch_bounds__(256) k_mean_combine(const int32_t* __restrict__ seg_start, int nv,
                                                      const long long* __restrict__ pieces, float* __restrict__ mean) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const int b = seg_start[v], e = seg_start[v + 1];
  if (b == e) return;
  const int kb = b / MEAN_BLOCK, ke = (e - 1) / MEAN_BLOCK;
  if (kb == ke) return;  // finished by k_mean_pieces
  const long long* d = pieces + ((int64_t)kb * 2 + ((b % MEAN_BLOCK == 0) ? 0 : 1)) * 3;
  long long x = d[0], y = d[1], z = d[2];
  for (int k = kb + 1; k <= ke; ++k) {
    d = pieces + (int64_t)k * 2 * 3;
    x += d[0];
    y += d[1];
    z += d[2];
  }
  const double cnt = (double)(e - b);
  mean[3 * v] = tln_unfix20(x, cnt);
  mean[3 * v + 1] = tln_unfix20(y, cnt);
  mean[3 * v + 2] = tln_unfix20(z, cnt);
}

__global__ void __launch_bounds__(256) k_subtract_rows(const float* __restrict__ pos, const int32_t* __restrict__ indices,
                                                       const float* __restrict__ mean, int64_t rows, int cols,
                                                       float* __restrict__ dist) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const int v = indices[row];
  if (v < 0) return;  // rows without a vertex keep the raw position
  const int64_t p = row >> 2;
  float* d = dist + row * cols;
  d[0] = pos[3 * p] - mean[3 * v];
  d[1] = pos[3 * p + 1] - mean[3 * v + 1];
  d[2] = pos[3 * p + 2] - mean[3 * v + 2];
}

// ---------------------------------------------------------------------------------------
// K1 phase D (distribute path): vertex bins instead of a sort.
//   k_bins_alloc    one thread per vertex: its row count of this frame (from the slot counts of phase A) and a
//                   contiguous segment of the bin arrays (wave prefix sum + ONE atomic per wave on the cursor)
//   k_bins_scatter  one thread per row: vertex index (the `indices` output), the row's payload to
//                   segment start + rank, reset of the slot count by the row of rank 0
//   k_bins_mean     one wave per vertex: fixed-point position sum of its segment (int64: exact, hence
//                   order-independent) / count
// Rows without a vertex (key out of range, rejected by the capacity) go to the tail behind all segments.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bins_alloc(int32_t* __restrict__ ctr, const int32_t* __restrict__ vslot,
                                                    const uint32_t* __restrict__ slot_cnt, int32_t* __restrict__ vcnt,
                                                    int32_t* __restrict__ vstart) {
  const int nv = ctr[CTR_NV];
  const int lane = threadIdx.x & 63;
  const int stride = gridDim.x * blockDim.x;
  for (int v0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63); v0 < nv; v0 += stride) {   // whole waves stay together
    const int v = v0 + lane;
    const int c = v < nv ? (int)slot_cnt[vslot[v]] : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 0 && total > 0) base = atomicAdd(&ctr[CTR_CURSOR], total);
    base = __shfl(base, 0, 64);
    if (v < nv) {
      vcnt[v] = c;
      vstart[v] = base + incl - c;
    }
  }
}

__global__ void __launch_bounds__(256) k_bins_scatter(TableRef t, const int32_t* __restrict__ row_slot,
                                                      const int32_t* __restrict__ row_rank, int64_t rows,
                                                      const float* __restrict__ pos, const float* __restrict__ val,
                                                      int val_dim, const float* __restrict__ weights,
                                                      uint32_t* __restrict__ slot_cnt, const int32_t* __restrict__ vstart,
                                                      int32_t* __restrict__ indices,
                                                      TlnBinRec* __restrict__ bin_rec) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const int slot = row_slot[row];
  const int rank = row_rank[row];
  int v = -1;
  if (slot >= 0) {
    v = t.slots[slot].val;
    if (rank == 0) slot_cnt[slot] = 0;   // exactly one row per touched slot: the counts are zero again for the next frame
  }
  if (indices) indices[row] = v;
  const int64_t p = row >> 2;
  const float x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
  int dest;
  if (v >= 0) {
    dest = vstart[v] + rank;
  } else {
    atomicAdd(&t.ctr[CTR_OVERFLOW], 1);
    // behind all segments: CTR_CURSOR is final here (k_bins_alloc has completed)
    dest = t.ctr[CTR_CURSOR] + atomicAdd(&t.ctr[CTR_TAIL], 1);
  }
  bin_rec[dest].a = make_float4(x, y, z, val_dim == 1 ? val[p] : 0.f);
  bin_rec[dest].m = make_uint4(__float_as_uint(weights[row]), (uint32_t)row, (uint32_t)v, 0u);
}

// one WAVE per vertex: its rows are contiguous in the bins; positions summed in fixed point (int64: exact, any order)
__global__ void __launch_bounds__(256) k_bins_mean(const int32_t* __restrict__ ctr, const int32_t* __restrict__ vcnt,
                                                   const int32_t* __restrict__ vstart,
                                                   const TlnBinRec* __restrict__ bin_rec, float* __restrict__ mean) {
  const int nv = ctr[CTR_NV];
  const int lane = threadIdx.x & 63;
  const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  for (int v = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); v < nv; v += nwaves) {   // the host only has a bound of nv
    const int c = vcnt[v], st = vstart[v];
    long long sx = 0, sy = 0, sz = 0;
    for (int j = lane; j < c; j += 64) {
      const float4 q = bin_rec[st + j].a;
      sx += tln_fix20(q.x);
      sy += tln_fix20(q.y);
      sz += tln_fix20(q.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sx += __shfl_xor(sx, o, 64);
      sy += __shfl_xor(sy, o, 64);
      sz += __shfl_xor(sz, o, 64);
    }
    if (lane == 0) {
      const double cnt = (double)(c > 0 ? c : 1);
      mean[3 * v] = tln_unfix20(sx, cnt);
      mean[3 * v + 1] = tln_unfix20(sy, cnt);
      mean[3 * v + 2] = tln_unfix20(sz, cnt);
    }
  }
}


// ---------------------------------------------------------------------------------------
// K1, partitioned: the same distribute WITHOUT a global atomic per row.
// A scattered device-scope atomic costs ~43 ps of chip time whatever its scope (tools/micro/atomics.hip: 480k of them
// = 21 us; every add to ONE word another ~11 ns), and k_distribute_insert + k_bins_scatter need three per row when the
// points arrive shuffled.  Here the rows are first split by 8-13 bits of their key's hash into buckets of ~500 rows, and
// ONE workgroup owns a bucket: every per-row atomic (find-or-insert, row count, first touch, fixed-point position sums)
// is an LDS atomic on the bucket's own 1024-entry table; the global table sees one probe per DISTINCT key of the frame.
//   k_bk_split   block = a range of 256-2048 points (1024 on a 120k-point frame; a thread per point up to 1024
//                threads): simplex arithmetic, LDS histogram of the buckets, then the 32-byte records
//                {x, y, z, value, weight, row, key} written bucket by bucket into the block's own region (+ the block's
//                bucket offsets, stored bucket-major); also the per-row weights and, when asked for, the [4N,5] rows
//   k_bk_insert  block = a bucket: walks its runs (one per split block, ~points per block / 128 records each; as many
//                threads share a run as fit the workgroup), LDS find-or-insert with the smallest row per key; then one
//                global probe per distinct key (a new key: one plain 16-byte store), the bit of the first-touch row of
//                every key without a vertex in a bit mask over the rows, the bucket's row count
//   k_bk_prefix  one workgroup: set bits of the mask before every 128 rows = the first-touch numbering a scan over the
//                rows in order would hand out; the counters, also into the host's mapped words
//   k_bk_place   block = a bucket: LDS table again, now with row counts and position sums; one global lookup per
//                distinct key -> vertex; the bucket's rows go to the bin range [rows of the buckets before, +own rows),
//                one segment per vertex (the layout k_pool_bins reads), rows without a vertex behind them; indices[row]
// The order of the rows inside a segment is arbitrary, as before; every result computed from the bins is
// order-independent.  Vertices without rows in this frame are told apart by vstamp (nobody visits them).
// ---------------------------------------------------------------------------------------
// bucket of a key = the top bits of its home slot: the slot ranges of the buckets are disjoint (group-local probing)
__device__ __forceinline__ uint32_t bk_bucket(uint64_t K, uint64_t slot_mask, int shift) {
  return K == TLN_KEY_EMPTY ? 0u : (uint32_t)((tln_mix64(K) & slot_mask) >> shift);
}

// MEASURED AND NOT KEPT (round 4, -DTLN_K1_NT=1): the streams of K1 — row records read once by each bucket kernel, bin
// records written once — as NON-TEMPORAL accesses, so that the frame's packed points (1.9 MB, gathered by row id, every
// line wanted ~4 times per XCD) would stay in an XCD's 4 MB L2.  k_bk_place 171.8 -> 211.6 us per batch of eight frames,
// k_bk_insert 59.3 -> 66.6, fetched bytes unchanged (555 against 565 MB): the runs are 64 bytes, two to four lanes share a
// line, and the non-temporal loads gave that sharing up.
#ifndef TLN_K1_NT
#define TLN_K1_NT 0
#endif
typedef unsigned int bk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 bk_load_rec(const uint4* p) {
#if TLN_K1_NT & 1   // (bit 0: the record loads, bit 1: the bin stores)
  const bk_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bk_u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void bk_store_stream(uint4* p, const uint4& v) {
#if TLN_K1_NT & 2
  bk_u32x4 w;
  w.x = v.x;
  w.y = v.y;
  w.z = v.z;
  w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<bk_u32x4*>(p));
#else
  *p = v;
#endif
}

// exclusive prefix sum over the threads of a block (whole waves, at most 16); *total = sum of all (valid after the call
// for every thread)
__device__ __forceinline__ uint32_t bk_block_scan(uint32_t v, uint32_t* wtmp /* [16] shared */, uint32_t* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  __syncthreads();   // wtmp may still be read from an earlier scan
  if (lane == 63) wtmp[wid] = incl;
  __syncthreads();
  uint32_t pre = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    if (w < wid) pre += wtmp[w];
    tot += wtmp[w];
  }
  *total = tot;
  return pre + incl - v;
}

// One frame of one lattice as the four kernels see it.  The kernels take up to TLN_BK_MAXJOBS of them (blockIdx.y = job):
// a host that steps several sequences in lock-step (the frame program's group mode, bench.py's timed mode) distributes
// the frames of all of them with FOUR launches instead of four per frame — every one of these kernels is a chain of
// dependent memory round trips (~4.5 us from launch to the first loaded byte), and a 120k-point frame alone does not
// fill the chip (117 split blocks; the buckets of one frame in one round of workgroups).
#define TLN_BK_MAXJOBS 8
struct BkJob {
  const float* pos;
  const float* val;
  float4* posv;              // [n] {x, y, z, value}: written by k_bk_split, gathered by k_bk_place (one 16-byte load per row)
  float* weights;
  float* dist;
  uint4* rec;                // 16-byte row records {weight bits, row, key lo, key hi}, grouped by (split block, bucket)
  uint32_t* off;
  uint32_t* first_bits;
  uint32_t* bucket_rows;
  uint32_t* bits_pre;
  int32_t* ctr;
  int32_t* host_ctr;
  int32_t* vstart;
  int32_t* vcnt;
  int32_t* vstamp;
  int32_t* indices;
  int32_t* vkeys;
  int32_t* vslot;
  float* mean;
  TlnBinRec* bin_rec;
  TableRef t;
  uint64_t slot_mask;
  int64_t n, rows, rpb;
  float s0, s1, s2;
  int val_dim, ppb, B, nblk, shift, nr_words, vold, capacity, stamp, stage;
};
struct BkJobs {
  BkJob j[TLN_BK_MAXJOBS];
  int xcd;   // 1: the frames of a batch are dealt to the XCDs (bk_block)
};

// frames of a batch <-> XCDs: common.h's tln_xcd_block — the frame's packed points (gathered by row id), its indices
// (scattered 4-byte stores) and the lines of its buckets meet in one L2
__device__ __forceinline__ void bk_block(const BkJobs& jobs, int& job, int& bx) { tln_xcd_block(jobs.xcd, bx, job); }

__global__ void __launch_bounds__(1024) k_bk_split(const BkJobs jobs) {
  int job, bx;
  bk_block(jobs, job, bx);
  const BkJob& J = jobs.j[job];
  const int nblk = J.nblk;
  if (bx >= nblk) return;   // (the grid is sized for the job with the most split blocks)
  extern __shared__ uint32_t bk_hist[];   // [B] bucket counts of this block, then the write cursors
  __shared__ uint32_t wtmp[16];
  const int tid = threadIdx.x, T = blockDim.x;   // T: 256, 512 or 1024 (a power of two, as B)
  const int B = J.B, ppb = J.ppb, val_dim = J.val_dim;
  const float* __restrict__ pos = J.pos;
  const float* __restrict__ val = J.val;
  const float s0 = J.s0, s1 = J.s1, s2 = J.s2;
  const uint64_t slot_mask = J.slot_mask;
  const int shift = J.shift;
  const int64_t n = J.n;
  // what k_bk_insert accumulates into: the bit mask of the first-touch rows, their number
  for (int i = bx * T + tid; i < J.nr_words; i += nblk * T) J.first_bits[i] = 0u;
  if (bx == 0 && tid == 0) J.ctr[CTR_OVERFLOW] = 0;   // accumulated by k_bk_place
  for (int i = tid; i < B; i += T) bk_hist[i] = 0;
  __syncthreads();
  const int64_t p0 = (int64_t)bx * ppb;
  const int64_t p1 = p0 + ppb < n ? p0 + ppb : n;
  float* __restrict__ weights = J.weights;
  float* __restrict__ dist = J.dist;
  for (int64_t p = p0 + tid; p < p1; p += T) {
    const float x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
    int rem0[4], rank[4];
    float bary[4];
    point_simplex(x, y, z, s0, s1, s2, rem0, rank, bary);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int k0, k1, k2;
      vertex_key(rem0, rank, r, k0, k1, k2);
      const uint64_t K = tln_key_in_range(k0, k1, k2) ? tln_pack_key(k0, k1, k2) : TLN_KEY_EMPTY;
      atomicAdd(&bk_hist[bk_bucket(K, slot_mask, shift)], 1u);
    }
    *reinterpret_cast<float4*>(weights + 4 * p) = make_float4(bary[0], bary[1], bary[2], bary[3]);
    // the point as ONE 16-byte record for k_bk_place's gather by row id (position 12 B + value 4 B from two arrays were
    // two cache lines per row, and with eight frames per launch their 15 MB no longer sat in an XCD's L2: PMC FETCH_SIZE)
    J.posv[p] = make_float4(x, y, z, val_dim ? val[p] : 0.0f);
    if (dist) {   // the [4N, 3 + val_dim + 1] rows (val_dim <= 1 on this path): raw position, value, weight
      const int cols = 3 + val_dim + 1;
      float* d = dist + 4 * p * cols;
      const float v = val_dim ? val[p] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[r * cols] = x;
        d[r * cols + 1] = y;
        d[r * cols + 2] = z;
        if (val_dim) d[r * cols + 3] = v;
        d[r * cols + 3 + val_dim] = bary[r];
      }
    }
  }
  __syncthreads();
  // exclusive scan of the B counts: a thread owns B/T consecutive buckets (one, for B < T).  off is bucket-major ([B + 1][nblk]) so that
  // a bucket's workgroup reads its runs with contiguous loads.
  uint32_t* __restrict__ off = J.off;
  const int per = B >= T ? B / T : (tid < B ? 1 : 0);
  uint32_t mine = 0;
  for (int k = 0; k < per; ++k) mine += bk_hist[tid * per + k];
  uint32_t total;
  uint32_t run = bk_block_scan(mine, wtmp, &total);
  for (int k = 0; k < per; ++k) {
    const uint32_t c = bk_hist[tid * per + k];
    bk_hist[tid * per + k] = run;
    off[(size_t)(tid * per + k) * nblk + bx] = run;
    run += c;
  }
  if (tid == 0) off[(size_t)B * nblk + bx] = total;
  __syncthreads();
  // the 16-byte records: weight, row, key.  The position / value of a row are re-read by k_bk_place from the frame's
  // own arrays (1.9 MB, cache resident) — carrying them along made the record 32 bytes, written once and pulled twice.
  // They go to the block's region grouped by bucket THROUGH LDS (J.stage): scattered 16-byte stores to ~1000 runs per
  // block reached memory as partial lines (2.3 x the bytes, PMC WRITE_SIZE); the staged image leaves as whole lines.
  uint4* region = J.rec + (size_t)bx * (4 * (size_t)ppb);
  uint4* stage = reinterpret_cast<uint4*>(bk_hist + ((B + 3) & ~3));
  uint4* dst = J.stage ? stage : region;
  for (int64_t p = p0 + tid; p < p1; p += T) {
    const float x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
    int rem0[4], rank[4];
    float bary[4];
    point_simplex(x, y, z, s0, s1, s2, rem0, rank, bary);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int k0, k1, k2;
      vertex_key(rem0, rank, r, k0, k1, k2);
      const uint64_t K = tln_key_in_range(k0, k1, k2) ? tln_pack_key(k0, k1, k2) : TLN_KEY_EMPTY;
      const uint32_t at = atomicAdd(&bk_hist[bk_bucket(K, slot_mask, shift)], 1u);
      dst[at] = make_uint4(__float_as_uint(bary[r]), (uint32_t)(4 * p + r), (uint32_t)K, (uint32_t)(K >> 32));
    }
  }
  if (J.stage) {
    __syncthreads();
    const int nrec = (int)(4 * (p1 - p0));
    for (int i = tid; i < nrec; i += T) bk_store_stream(&region[i], stage[i]);
  }
}

// LDS find-or-insert of key K in a bucket's table: entry, or -1 when the table is full
__device__ __forceinline__ int bk_lds_insert(unsigned long long* hk, uint64_t K) {
  uint32_t h = (uint32_t)(tln_mix64(K) >> 24) & (TLN_BK_HT - 1);
  for (int probe = 0; probe < TLN_BK_HT; ++probe) {
    unsigned long long cur = hk[h];
    if (cur == TLN_KEY_EMPTY) {
      cur = atomicCAS(&hk[h], (unsigned long long)TLN_KEY_EMPTY, (unsigned long long)K);
      if (cur == TLN_KEY_EMPTY) cur = K;
    }
    if (cur == K) return (int)h;
    h = (h + 1) & (TLN_BK_HT - 1);
  }
  return -1;
}
__device__ __forceinline__ int bk_lds_find(const unsigned long long* hk, uint64_t K) {
  uint32_t h = (uint32_t)(tln_mix64(K) >> 24) & (TLN_BK_HT - 1);
  for (int probe = 0; probe < TLN_BK_HT; ++probe) {
    const unsigned long long cur = hk[h];
    if (cur == K) return (int)h;
    if (cur == TLN_KEY_EMPTY) return -1;
    h = (h + 1) & (TLN_BK_HT - 1);
  }
  return -1;
}

// the rows of bucket b that thread tid walks.  Run j = the bucket's records inside split block j's region (contiguous;
// at most TLN_BK_THREADS split blocks); 2^sh threads share a run (as many as fit the workgroup), thread q of them takes
// its rows q, q + 2^sh, ...: neighbouring threads read neighbouring records
#define BK_KEEP 1   // rows a thread keeps in registers between its two sweeps (a thread has one row, seldom two, now that the
                    // threads of a run share it; with three k_bk_place needed 84 VGPRs: three workgroups per CU, and a
                    // frame's 1024 buckets ran in two rounds)
struct BkRuns {
  const uint4* at;
  int total, step;
};
__device__ __forceinline__ void bk_runs_of(const uint32_t* __restrict__ off, const uint4* __restrict__ rec, int nblk,
                                           int64_t rpb, int b, BkRuns& rn) {
  int sh = 0;
  while ((nblk << (sh + 1)) <= TLN_BK_THREADS) ++sh;
  const int j = threadIdx.x >> sh, q = threadIdx.x & ((1 << sh) - 1);
  uint32_t s0 = 0, e0 = 0;
  if (j < nblk) {
    s0 = off[(size_t)b * nblk + j];
    e0 = off[(size_t)(b + 1) * nblk + j];
  }
  const int len = (int)(e0 - s0);
  rn.at = rec + (size_t)j * rpb + s0 + q;
  rn.step = 1 << sh;
  rn.total = len > q ? (len - q + rn.step - 1) >> sh : 0;
}
__device__ __forceinline__ const uint4* bk_rec_of(const BkRuns& rn, int k) { return rn.at + k * rn.step; }


__global__ void __launch_bounds__(TLN_BK_THREADS) k_bk_insert(const BkJobs jobs) {
  int job, bx;
  bk_block(jobs, job, bx);
  const BkJob& J = jobs.j[job];
  if (bx >= J.B) return;
  __shared__ unsigned long long hk[TLN_BK_HT];
  __shared__ uint32_t htouch[TLN_BK_HT];
  __shared__ uint32_t claimed[TLN_BK_HT];   // slots this workgroup has taken in this launch (open addressing, 0 = free)
  __shared__ uint32_t wtmp[16];
  const int tid = threadIdx.x, b = bx;
  const TableRef t = J.t;
  for (int i = tid; i < TLN_BK_HT; i += TLN_BK_THREADS) {
    hk[i] = TLN_KEY_EMPTY;
    htouch[i] = 0xFFFFFFFFu;
    claimed[i] = 0u;
  }
  // the thread's run: every offset load is issued before the first record load, every
  // record load before the first LDS operation — the kernel is a chain of dependent memory round trips otherwise
  BkRuns rn;
  bk_runs_of(J.off, J.rec, J.nblk, J.rpb, b, rn);
  uint4 kb[BK_KEEP];
#pragma unroll
  for (int k = 0; k < BK_KEEP; ++k)
    if (k < rn.total) kb[k] = bk_load_rec(bk_rec_of(rn, k));
  __syncthreads();
  auto touch_row = [&](const uint4& bb) {
    const uint64_t K = ((uint64_t)bb.w << 32) | bb.z;
    if (K == TLN_KEY_EMPTY) return;
    const int he = bk_lds_insert(hk, K);
    if (he < 0) atomicAdd(&t.ctr[CTR_BUCKET_FULL], 1);   // more distinct keys in one bucket than its table holds
    else atomicMin(&htouch[he], bb.y);
  };
#pragma unroll
  for (int k = 0; k < BK_KEEP; ++k)
    if (k < rn.total) touch_row(kb[k]);
  for (int k = BK_KEEP; k < rn.total; ++k) touch_row(bk_load_rec(bk_rec_of(rn, k)));
  const uint32_t nrows = (uint32_t)rn.total;
  uint32_t R;
  bk_block_scan(nrows, wtmp, &R);   // (barriers: the table is complete behind it)
  if (tid == 0) J.bucket_rows[b] = R;
  uint32_t* __restrict__ first_bits = J.first_bits;
  // The distinct keys of the bucket, one global probe each.  Only this workgroup writes the bucket's slot range in
  // this launch, so a key is entered with ONE plain 16-byte store; two of its threads heading for the same empty
  // slot settle it in LDS (`claimed`), the loser moves on along its group.
  for (int e = tid; e < TLN_BK_HT; e += TLN_BK_THREADS) {
    const unsigned long long K = hk[e];
    if (K == TLN_KEY_EMPTY) continue;
    const uint32_t touch = htouch[e];
    uint64_t slot = tln_mix64(K) & t.mask;
    bool found = false, numbered = false;
    for (int probe = 0; probe < TLN_MAX_PROBES && !found; ++probe) {
      unsigned long long cur;
      int val;
      uint32_t tch;
      load_slot(t, slot, cur, val, tch);
      if (cur == K) {            // there from an earlier frame (keys are unique inside the frame's table)
        numbered = val >= 0;
        if (!numbered) t.slots[slot].touch = touch;   // was rejected by the capacity before: tried again
        found = true;
        break;
      }
      if (cur == TLN_KEY_EMPTY) {
        // free in memory; free among this launch's claims too?
        const uint32_t tag = (uint32_t)slot + 1u;
        uint32_t h = (uint32_t)(slot * 0x9E3779B1u) & (TLN_BK_HT - 1);
        bool mine = false;
        for (int q = 0; q < TLN_BK_HT; ++q) {
          uint32_t c = claimed[h];
          if (c == 0u) {
            c = atomicCAS(&claimed[h], 0u, tag);
            if (c == 0u) {
              mine = true;
              break;
            }
          }
          if (c == tag) break;   // another key of this bucket took the slot a moment ago
          h = (h + 1) & (TLN_BK_HT - 1);
        }
        if (mine) {
          ulonglong2 rec16;
          rec16.x = K;
          rec16.y = 0x00000000FFFFFFFFull | ((unsigned long long)touch << 32);   // vertex -1, first-touch row
          *reinterpret_cast<ulonglong2*>(&t.slots[slot]) = rec16;
          found = true;
          break;
        }
      }
      slot = tln_next_slot(slot);
    }
    if (!found) {
      atomicAdd(&t.ctr[CTR_PROBE_FAIL], 1);
    } else if (!numbered) {
      atomicOr(&first_bits[touch >> 5], 1u << (touch & 31));   // k_bk_place numbers the keys by these bits
    }
  }
}

// The first-touch rows of the frame as a bit mask -> for every uint4 of the mask the number of set bits before it (one
// workgroup per frame: 15k words on a 120k-point frame), and the counters the host fetches while k_bk_place runs.  A
// key's vertex index is then: vertices before the frame + set bits before its own first-touch row (what a scan over the
// rows in order would hand out).  The occupancy count takes every key without a vertex as newly entered: keys turned
// away by the capacity in an earlier frame count again, which only brings the next table rebuild (exact again) forward.
__global__ void __launch_bounds__(TLN_BK_THREADS) k_bk_prefix(const BkJobs jobs) {
  const BkJob& J = jobs.j[blockIdx.y];
  __shared__ uint32_t wtmp[16];
  // a thread owns `per` consecutive uint4s of the mask: one pass over them for the sum, one block scan, one pass to write
  const int nq = J.nr_words >> 2;
  const int per = (nq + TLN_BK_THREADS - 1) / TLN_BK_THREADS;
  const uint4* q4 = reinterpret_cast<const uint4*>(J.first_bits);
  uint32_t* __restrict__ bits_pre = J.bits_pre;
  const int i0 = threadIdx.x * per, i1 = i0 + per < nq ? i0 + per : nq;
  uint32_t mine = 0;
  for (int i = i0; i < i1; ++i) {
    const uint4 q = q4[i];
    mine += (uint32_t)(__popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w));
  }
  uint32_t carry;
  uint32_t run = bk_block_scan(mine, wtmp, &carry);
  for (int i = i0; i < i1; ++i) {
    const uint4 q = q4[i];
    bits_pre[i] = run;
    run += (uint32_t)(__popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w));
  }
  if (threadIdx.x == 0) {
    int32_t* ctr = J.ctr;
    // a bucket's LDS table overflowed (k_bk_insert): nothing of this frame is numbered — k_bk_place returns at once,
    // the host redoes the frame with the per-row-atomic kernels (the keys already entered stay, un-numbered, with
    // their first-touch rows: exactly what those kernels expect to find)
    const bool full = ctr[CTR_BUCKET_FULL] != 0;
    const long long vnew = (long long)J.vold + (full ? 0u : carry);
    ctr[CTR_VOLD] = J.vold;
    ctr[CTR_NEW] = full ? 0 : (int)carry;
    ctr[CTR_NV] = (int)(vnew < J.capacity ? vnew : J.capacity);
    ctr[CTR_OCCUPIED] += (int)carry;
    // the host's copy (pinned, mapped): written from here instead of a copy kernel between this launch and k_bk_place
    for (int i = 0; i < CTR_COUNT; ++i) J.host_ctr[i] = ctr[i];
  }
}

__global__ void __launch_bounds__(TLN_BK_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_bk_place(const BkJobs jobs) {
  int job, bx;
  bk_block(jobs, job, bx);
  const BkJob& J = jobs.j[job];
  if (bx >= J.B) return;
  if (J.ctr[CTR_BUCKET_FULL] != 0) return;             // (uniform) the frame is redone by the per-row-atomic kernels
  __shared__ unsigned long long hk[TLN_BK_HT];
  __shared__ uint32_t hcnt[TLN_BK_HT];                 // rows per entry, then the entry's write cursor
  extern __shared__ unsigned long long bk_dyn[];       // (dynamic: the block needs more than 64 KB of LDS in all)
  unsigned long long (*hsum)[TLN_BK_HT] = reinterpret_cast<unsigned long long (*)[TLN_BK_HT]>(bk_dyn);   // [3][HT] fixed-point position sums; afterwards reused:
  int* hstart = reinterpret_cast<int*>(&hsum[0][0]);   //   [HT] first bin position of the entry's vertex
  int* hv = hstart + TLN_BK_HT;                        //   [HT] vertex of the entry (-1: none)
  // ... and, behind them, the first BK_STAGE bin records of the bucket: its bin range [bin0, bin0 + R) is contiguous, so
  // the records are collected here and leave as whole lines (round 3 wrote each 32-byte record where its lane happened
  // to be: 15 MB of scattered half-lines per frame); a bucket with more rows writes the rest directly
  constexpr int BK_STAGE = (int)((3 * TLN_BK_HT * sizeof(unsigned long long) - 2 * TLN_BK_HT * sizeof(int)) / sizeof(TlnBinRec));
  TlnBinRec* bstage = reinterpret_cast<TlnBinRec*>(hv + TLN_BK_HT);
  __shared__ uint32_t wtmp[16];
  __shared__ uint32_t s_tail;
  const int tid = threadIdx.x, b = bx;
  const TableRef t = J.t;
  for (int i = tid; i < TLN_BK_HT; i += TLN_BK_THREADS) {
    hk[i] = TLN_KEY_EMPTY;
    hcnt[i] = 0;
    hsum[0][i] = hsum[1][i] = hsum[2][i] = 0ull;
  }
  if (tid == 0) s_tail = 0;
  // the thread's rows (bk_runs_of: offsets first, then every record, then the rows' positions, then the LDS work) and,
  // meanwhile, the bins of this bucket: behind the rows of all buckets before it
  BkRuns rn;
  bk_runs_of(J.off, J.rec, J.nblk, J.rpb, b, rn);
  uint32_t before = 0;
  for (int i = tid; i < b; i += TLN_BK_THREADS) before += J.bucket_rows[i];
  float4 ka[BK_KEEP];
  uint4 kb[BK_KEEP];
#pragma unroll
  for (int k = 0; k < BK_KEEP; ++k)
    if (k < rn.total) kb[k] = bk_load_rec(bk_rec_of(rn, k));
  const float4* __restrict__ posv = J.posv;
#ifdef TLN_K1_KO_GATHER   // measurement builds only (results wrong): where do k_bk_place's HBM-side bytes come from?
  auto payload = [&](uint32_t row) { return make_float4(1.f, 2.f, 3.f, (float)(row & 7)); };
#else
  auto payload = [&](uint32_t row) { return posv[row >> 2]; };   // (x, y, z, value): one 16-byte load
#endif
#pragma unroll
  for (int k = 0; k < BK_KEEP; ++k)
    if (k < rn.total) ka[k] = payload(kb[k].y);
  uint32_t bin0;
  bk_block_scan(before, wtmp, &bin0);   // (barriers: the table is initialised behind it)
  const uint32_t R = J.bucket_rows[b];
  // sweep 1: rows per key, position sums
  auto count_row = [&](const float4& a, const uint4& bb) {
    const uint64_t K = ((uint64_t)bb.w << 32) | bb.z;
    if (K == TLN_KEY_EMPTY) return;
    const int he = bk_lds_insert(hk, K);
    if (he < 0) return;   // (reported by k_bk_insert)
    atomicAdd(&hcnt[he], 1u);
    atomicAdd(&hsum[0][he], (unsigned long long)tln_fix20(a.x));
    atomicAdd(&hsum[1][he], (unsigned long long)tln_fix20(a.y));
    atomicAdd(&hsum[2][he], (unsigned long long)tln_fix20(a.z));
  };
#pragma unroll
  for (int k = 0; k < BK_KEEP; ++k)
    if (k < rn.total) count_row(ka[k], kb[k]);
  for (int k = BK_KEEP; k < rn.total; ++k) {
    const uint4 bb = bk_load_rec(bk_rec_of(rn, k));
    count_row(payload(bb.y), bb);
  }
  __syncthreads();
  // the distinct keys: vertex (one global lookup), segment
  const uint32_t* __restrict__ first_bits = J.first_bits;
  const uint32_t* __restrict__ bits_pre = J.bits_pre;
  const int vold = J.vold, capacity = J.capacity;
  float* __restrict__ mean = J.mean;
  constexpr int EPT = TLN_BK_HT / TLN_BK_THREADS;
  int v[EPT];
  uint32_t c[EPT];
  uint32_t placed = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid * EPT + k;
    c[k] = hcnt[e];
    v[k] = -1;
    if (c[k]) {
      const unsigned long long K = hk[e];
      uint64_t slot = tln_mix64(K) & t.mask;
      bool found = false;
      uint32_t tch = 0;
      for (int probe = 0; probe < TLN_MAX_PROBES; ++probe) {
        unsigned long long cur;
        load_slot(t, slot, cur, v[k], tch);
        if (cur == K) {
          found = true;
          break;
        }
        if (cur == TLN_KEY_EMPTY) break;
        slot = tln_next_slot(slot);
      }
      if (!found) v[k] = -1;
      if (found && v[k] < 0) {
        // a key without a vertex: its number = vertices before + first-touch rows before its own first-touch row
        const uint32_t w = tch >> 5;
        const uint4 q = reinterpret_cast<const uint4*>(first_bits)[w >> 2];
        const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
        uint32_t rank = bits_pre[w >> 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if ((uint32_t)i < (w & 3u)) rank += (uint32_t)__popc(qw[i]);
          if ((uint32_t)i == (w & 3u)) rank += (uint32_t)__popc(qw[i] & ((1u << (tch & 31)) - 1u));
        }
        const long long vn = (long long)vold + rank;
        if (vn < capacity) {
          v[k] = (int)vn;
          t.slots[slot].val = v[k];
          J.vslot[vn] = (int32_t)slot;
          int k0, k1, k2;
          tln_unpack_key(K, k0, k1, k2);
          *reinterpret_cast<int4*>(J.vkeys + 4 * vn) = make_int4(k0, k1, k2, -(k0 + k1 + k2));
        } else {
          t.slots[slot].touch = 0xFFFFFFFFu;  // stays un-numbered; may be retried by a later insertion
        }
      }
    }
    if (v[k] >= 0) {
      placed += c[k];
      // the local mean needs the sums and the vertex only: written here, so that the sums need not live in registers
      // across the scan below (64 VGPRs = four workgroups per CU: the 1024 buckets of a frame in ONE round)
      const double cnt = (double)c[k];
#ifndef TLN_K1_KO_VERTEX
      mean[3 * v[k]] = tln_unfix20((long long)hsum[0][e], cnt);
      mean[3 * v[k] + 1] = tln_unfix20((long long)hsum[1][e], cnt);
      mean[3 * v[k] + 2] = tln_unfix20((long long)hsum[2][e], cnt);
#endif
    }
  }
  uint32_t P;
  uint32_t st = bk_block_scan(placed, wtmp, &P);   // (barriers: every sum has been read, hsum may be overwritten)
  if (tid == 0 && R > P) atomicAdd(&t.ctr[CTR_OVERFLOW], (int)(R - P));
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid * EPT + k;
    hcnt[e] = 0;
    hv[e] = v[k];
    hstart[e] = (int)(bin0 + st);
    if (v[k] >= 0) {
      const int vv = v[k];
#ifndef TLN_K1_KO_VERTEX
      J.vstart[vv] = (int)(bin0 + st);
      J.vcnt[vv] = (int)c[k];
      J.vstamp[vv] = J.stamp;
#endif
      st += c[k];
    }
  }
  __syncthreads();
  // sweep 2: the rows move to their segments (any order inside), rows without a vertex behind the bucket's segments
  TlnBinRec* __restrict__ bin_rec = J.bin_rec;
  int32_t* __restrict__ indices = J.indices;
  auto place_row = [&](const float4& a, const uint4& bb) {
    const uint64_t K = ((uint64_t)bb.w << 32) | bb.z;
    int he = -1, vv = -1;
    if (K != TLN_KEY_EMPTY) he = bk_lds_find(hk, K);
    if (he >= 0) vv = hv[he];
    uint32_t dest;
    if (vv >= 0) dest = (uint32_t)hstart[he] + atomicAdd(&hcnt[he], 1u);
    else dest = bin0 + P + atomicAdd(&s_tail, 1u);
    const uint32_t local = dest - bin0;
    TlnBinRec* out = local < (uint32_t)BK_STAGE ? bstage + local : bin_rec + dest;
    out->a = a;   // (both halves of the 32-byte record together)
    out->m = make_uint4(bb.x, bb.y, (uint32_t)vv, 0u);
#ifndef TLN_K1_KO_INDICES
    if (indices) indices[bb.y] = vv;   // (NULL: nobody slices this frame — a scattered 4-byte store per row saved)
#endif
  };
#pragma unroll
  for (int k = 0; k < BK_KEEP; ++k)
    if (k < rn.total) place_row(ka[k], kb[k]);
  for (int k = BK_KEEP; k < rn.total; ++k) {
    const uint4 bb = bk_load_rec(bk_rec_of(rn, k));
    place_row(payload(bb.y), bb);
  }
  __syncthreads();
  // the staged records, as consecutive 16-byte pieces: whole lines
  {
    const int nst = (int)(R < (uint32_t)BK_STAGE ? R : (uint32_t)BK_STAGE);
    const uint4* src = reinterpret_cast<const uint4*>(bstage);
    uint4* dst = reinterpret_cast<uint4*>(bin_rec + bin0);
#ifndef TLN_K1_KO_BINS
    for (int i = tid; i < 2 * nst; i += TLN_BK_THREADS) bk_store_stream(&dst[i], src[i]);
#endif
  }
}

// accessors for pool.hip: the bins of the last distribute, if they describe (d_distributed, rows)
bool tln_lat_bins(const tln_lattice* l, const float* d_distributed, int64_t rows, TlnBins* out) {
  if (!l || l->bins_rows != rows || rows <= 0 || l->bins_dist != d_distributed || l->dist_val_dim != 1) return false;
  out->rec = l->bin_rec;
  out->vstart = l->vstart;
  out->vcnt = l->vcnt;
  out->mean = l->mean;
  out->ctr = l->d_ctr;
  out->weights = l->bins_weights;
  out->subtract = l->bins_subtract;
  out->vstamp = l->bins_stamped ? l->vstamp : nullptr;
  out->stamp = l->bins_stamp;
  return true;
}

extern "C" int64_t tln_lattice_bucket_fallbacks(const tln_lattice_t* l) { return l ? l->bucket_fallbacks : -1; }

extern "C" int tln_lattice_drop_bins(tln_lattice_t* l) {
  TLN_REQUIRE(l, "null lattice");
  l->bins_rows = -1;
  return TLN_OK;
}

static void publish_counts(tln_lattice* l) {
  set_vertices(l, l->h_ctr[CTR_NV]);
  l->occupied = l->h_ctr[CTR_OCCUPIED];
}

// which K1: the partitioned kernels (default) or the per-row-atomic ones (tln_options.k1_legacy of the handle, or env
// TLN_K1_LEGACY=1 read once; also taken for val_dim > 1 and for frames beyond 4M rows)
static bool k1_partitioned(const tln_lattice* l) {
  static const bool env_legacy = [] {
    const char* e = getenv("TLN_K1_LEGACY");
    return e && e[0] == '1';
  }();
  return !env_legacy && !l->opt.k1_legacy;
}
const tln_options& tln_lat_options(const tln_lattice* l) { return l->opt; }
extern "C" int tln_lattice_set_options(tln_lattice_t* l, const tln_options* opt) {
  TLN_REQUIRE(l, "null lattice");
  tln_options d;
  tln_options_init(&d);
  l->opt = opt ? *opt : d;
  return TLN_OK;
}

// ---- host side of K1 -------------------------------------------------------------------------------------------------
// what every first half does before its kernels: an abandoned first half settled, workspaces, slot table
static int distribute_prepare(tln_lattice* l, int64_t n, hipStream_t s) {
  if (l->dist_pending) {
    // an abandoned first half (its caller failed in between): its kernels have numbered vertices on the device, so the
    // host's counts must follow before anything else is inserted
    TLN_HIP(hipEventSynchronize(l->ctr_wait ? l->ctr_wait : l->ctr_event));
    l->dist_pending = false;
    publish_counts(l);
    l->bins_rows = -1;
    TLN_REQUIRE(l->h_ctr[CTR_PROBE_FAIL] == 0, "hash probing failed for %d rows (table too full)", l->h_ctr[CTR_PROBE_FAIL]);
  }
  const int64_t rows = 4 * n;
  int rc = ensure_rows(l, rows);
  if (rc) return rc;
  rc = ensure_slots(l, rows, s);
  if (rc) return rc;
  l->bins_rows = -1;
  l->csr_rows = -1;
  return TLN_OK;
}

static bool bk_eligible(const tln_lattice* l, int64_t n, int val_dim) {
  return k1_partitioned(l) && val_dim <= 1 && 4 * n <= (int64_t)TLN_BK_MAXB * TLN_BK_ROWS;
}

// the bucket geometry of a frame and everything its four kernels address; *split_t = threads of its split blocks
static int bk_fill_job(tln_lattice* l, const float* d_positions, const float* d_values, int64_t n, int val_dim,
                       float* d_distributed, int32_t* d_indices, float* d_weights, BkJob& J, int* split_t) {
  const int64_t rows = 4 * n;
  static const int env_rows = getenv("TLN_BK_ROWS") ? atoi(getenv("TLN_BK_ROWS")) : 0;   // measurement overrides
  static const int env_ppb = getenv("TLN_BK_PPB") ? atoi(getenv("TLN_BK_PPB")) : 0;
  const int want_rows = l->opt.k1_bucket_rows > 0 ? l->opt.k1_bucket_rows : env_rows;
  const int bucket_rows = want_rows >= 128 && want_rows <= 65536 ? want_rows : TLN_BK_ROWS;
  int B = TLN_BK_MINB;
  while ((int64_t)B * bucket_rows < rows && B < l->bk_maxb) B <<= 1;
  // points per split block: a block's run inside a bucket is ~ppb / 128 records long, and the longer the runs, the
  // fewer partly used cache lines the bucket kernels pull; ~100 split blocks are kept at least
  int64_t ppb = 256;
  while (ppb < 2048 && n / (2 * ppb) >= 96) ppb *= 2;
  if (env_ppb >= 256 && env_ppb <= TLN_BK_MAX_PPB) ppb = env_ppb & ~255;
  if (tln_cdiv(n, ppb) > TLN_BK_SPLIT_BLOCKS) ppb = (tln_cdiv(n, TLN_BK_SPLIT_BLOCKS) + 255) & ~(int64_t)255;
  static const int env_st = getenv("TLN_BK_SPLIT_T") ? atoi(getenv("TLN_BK_SPLIT_T")) : 0;
  *split_t = env_st == 256 || env_st == 512 || env_st == 1024 ? env_st : (ppb >= 1024 ? 1024 : (ppb >= 512 ? 512 : 256));
  const int nblk = (int)tln_cdiv(n, ppb);
  const int64_t rpb = 4 * ppb;
  TLN_REQUIRE(B <= l->bk_maxb && (int64_t)nblk * rpb <= l->rec_cap && nblk <= TLN_BK_SPLIT_BLOCKS,
              "bucket geometry out of range (B %d, %d split blocks)", B, nblk);
  int slot_bits = 0, b_bits = 0;
  while ((1ll << slot_bits) < l->nslots) ++slot_bits;
  while ((1 << b_bits) < B) ++b_bits;
  TLN_REQUIRE((int64_t)B * TLN_SLOT_GROUP <= l->nslots, "more buckets (%d) than slot groups", B);
  ++l->bins_stamp;
  l->bins_stamped = true;
  J = BkJob{};
  J.pos = d_positions;
  J.val = d_values;
  J.posv = l->posv;
  J.weights = d_weights;
  J.dist = d_distributed;
  J.rec = l->rec;
  J.off = l->bk_off;
  J.first_bits = l->first_flag;
  J.bucket_rows = l->bucket_rows;
  J.bits_pre = l->bits_pre;
  J.ctr = l->d_ctr;
  J.host_ctr = l->h_ctr_dev;
  J.vstart = l->vstart;
  J.vcnt = l->vcnt;
  J.vstamp = l->vstamp;
  J.indices = d_indices;
  J.vkeys = l->vkeys;
  J.vslot = l->vslot;
  J.mean = l->mean;
  J.bin_rec = l->bin_rec;
  J.t = table_ref(l);
  J.slot_mask = J.t.mask;
  J.n = n;
  J.rows = rows;
  J.rpb = rpb;
  J.s0 = l->scale[0];
  J.s1 = l->scale[1];
  J.s2 = l->scale[2];
  J.val_dim = val_dim;
  J.ppb = (int)ppb;
  J.B = B;
  J.nblk = nblk;
  J.shift = slot_bits - b_bits;
  J.nr_words = (int)(tln_cdiv(rows, 128) * 4);   // bit mask of the rows, whole uint4s
  J.vold = (int)l->nr_vertices;
  J.capacity = (int)l->capacity;
  J.stamp = l->bins_stamp;
  J.stage = 0;   // decided for the whole launch (bk_launch)
  return TLN_OK;
}

// the four launches for 1..TLN_BK_MAXJOBS frames (blockIdx.y = frame); `ev` is recorded behind k_bk_prefix: the vertex
// counters of every frame are in the hosts' mapped words by then, k_bk_place runs behind the wait
static int bk_launch(BkJobs& jobs, int n, int split_t, hipEvent_t ev, hipStream_t s) {
  int maxblk = 0, maxB = 0;
  for (int i = 0; i < n; ++i) {
    if (jobs.j[i].nblk > maxblk) maxblk = jobs.j[i].nblk;
    if (jobs.j[i].B > maxB) maxB = jobs.j[i].B;
  }
  // the split blocks stage their records in LDS when the largest block's image fits beside the histogram
  int maxppb = 0;
  for (int i = 0; i < n; ++i)
    if (jobs.j[i].ppb > maxppb) maxppb = jobs.j[i].ppb;
  static const bool stage_off = getenv("TLN_BK_STAGE_OFF") != nullptr;
  size_t split_lds = (size_t)((maxB + 3) & ~3) * sizeof(uint32_t);
  const size_t staged = split_lds + (size_t)4 * maxppb * sizeof(uint4);
  const bool stage = !stage_off && staged <= 144 * 1024;
  if (stage) split_lds = staged;
  for (int i = 0; i < n; ++i) jobs.j[i].stage = stage ? 1 : 0;
  for (int i = n; i < TLN_BK_MAXJOBS; ++i) jobs.j[i] = jobs.j[0];   // (never indexed: the grids have n rows)
  static const bool xcd_off = getenv("TLN_K1_XCD") != nullptr && atoi(getenv("TLN_K1_XCD")) == 0;   // (measurement)
  jobs.xcd = (xcd_off || !tln_xcd_on()) ? 0 : 1;
  static thread_local TlnLdsAttr split_attr;
  TLN_HIP(tln_set_max_lds(split_attr, reinterpret_cast<const void*>(k_bk_split), (int)split_lds));
  hipLaunchKernelGGL(k_bk_split, dim3((unsigned)maxblk, (unsigned)n), dim3(split_t), split_lds, s, jobs);
  hipLaunchKernelGGL(k_bk_insert, dim3((unsigned)maxB, (unsigned)n), dim3(TLN_BK_THREADS), 0, s, jobs);
  hipLaunchKernelGGL(k_bk_prefix, dim3(1, (unsigned)n), dim3(TLN_BK_THREADS), 0, s, jobs);
  TLN_LAUNCH_CHECK();
  TLN_HIP(hipEventRecord(ev, s));
  const size_t place_lds = (size_t)3 * TLN_BK_HT * sizeof(unsigned long long);
  static thread_local TlnLdsAttr place_attr;
  TLN_HIP(tln_set_max_lds(place_attr, reinterpret_cast<const void*>(k_bk_place), (int)place_lds));
  hipLaunchKernelGGL(k_bk_place, dim3((unsigned)maxB, (unsigned)n), dim3(TLN_BK_THREADS), place_lds, s, jobs);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

static void distribute_remember(tln_lattice* l, const float* d_positions, int64_t n, int val_dim, int subtract_mean,
                                float* d_distributed, const int32_t* d_indices, const float* d_weights) {
  l->dist_pending = true;
  l->dist_pos = d_positions;
  l->dist_out = d_distributed;
  l->dist_idx = d_indices;
  l->dist_rows = 4 * n;
  l->dist_val_dim = val_dim;
  l->dist_subtract = subtract_mean;
  l->bins_weights = d_weights;
  l->dist_w = const_cast<float*>(d_weights);
}

// K1 by the per-row-atomic kernels (val_dim > 1, frames beyond the partitioned kernels' range, the legacy switch, and the
// fallback when a bucket's LDS table overflowed): insertion, numbering, the counters' way to the host, the bins
static int distribute_legacy_launch(tln_lattice* l, const float* d_positions, const float* d_values, int64_t n, int val_dim,
                                    int subtract_mean, float* d_distributed, int32_t* d_indices, float* d_weights,
                                    hipStream_t s) {
  const int64_t rows = 4 * n;
  TableRef t = table_ref(l);
  l->bins_stamped = false;
  hipLaunchKernelGGL(k_distribute_insert, dim3((unsigned)tln_cdiv(4 * n, 256)), dim3(256), 0, s, d_positions, d_values, n,
                     val_dim, l->scale[0], l->scale[1], l->scale[2], t, l->row_slot, d_weights, d_distributed,
                     l->slot_cnt, l->row_rank);
  TLN_LAUNCH_CHECK();
  int rc = number_new(l, rows, s);
  if (rc) return rc;
  TLN_HIP(hipMemcpyAsync(l->h_ctr, l->d_ctr, CTR_COUNT * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  TLN_HIP(hipEventRecord(l->ctr_event, s));
  // the bins: the host does not know the new vertex count yet; the kernels read it and stride over the vertices
  hipLaunchKernelGGL(k_bins_alloc, dim3(256), dim3(256), 0, s, l->d_ctr, l->vslot, l->slot_cnt, l->vcnt, l->vstart);
  hipLaunchKernelGGL(k_bins_scatter, dim3((unsigned)tln_cdiv(rows, 256)), dim3(256), 0, s, t, l->row_slot, l->row_rank, rows,
                     d_positions, d_values, val_dim, d_weights, l->slot_cnt, l->vstart, d_indices, l->bin_rec);
  if (subtract_mean)
    hipLaunchKernelGGL(k_bins_mean, dim3(8192), dim3(256), 0, s, l->d_ctr, l->vcnt, l->vstart, l->bin_rec, l->mean);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// first half: hash insertion, numbering, the bins; the vertex counters start their way to the host.  d_distributed may be NULL: the [4N, 3+val_dim+1] rows are then not produced
// (the pool of the same frame reads the bins).
extern "C" int tln_distribute_begin(tln_lattice_t* l, const float* d_positions, const float* d_values, int64_t n,
                                    int val_dim, int subtract_mean, float* d_distributed, int32_t* d_indices,
                                    float* d_weights, void* stream_) {
  TLN_REQUIRE(l && d_positions && d_weights, "null argument");
  TLN_REQUIRE(l->level == 0, "distribute works on the finest level");
  TLN_REQUIRE(n > 0 && 4 * n < (1ll << 31), "nr of points %lld out of range", (long long)n);
  TLN_REQUIRE(val_dim >= 0 && val_dim <= 1024 && (val_dim == 0 || d_values), "bad val_dim %d", val_dim);
  // d_indices may be NULL (a frame nobody slices) where the partitioned kernels run and no [4N, .] rows are asked for
  TLN_REQUIRE(d_indices || (bk_eligible(l, n, val_dim) && !d_distributed), "d_indices is required here");
  hipStream_t s = (hipStream_t)stream_;
  int rc = distribute_prepare(l, n, s);
  if (rc) return rc;
  if (!l->ctr_event) TLN_HIP(hipEventCreateWithFlags(&l->ctr_event, hipEventDisableTiming));
  l->ctr_wait = nullptr;
  if (bk_eligible(l, n, val_dim)) {
    // partitioned K1 (k_bk_*): four launches, no global atomic per row
    BkJobs jobs;
    int split_t = 0;
    rc = bk_fill_job(l, d_positions, d_values, n, val_dim, d_distributed, d_indices, d_weights, jobs.j[0], &split_t);
    if (rc) return rc;
    rc = bk_launch(jobs, 1, split_t, l->ctr_event, s);
    if (rc) return rc;
  } else {
    rc = distribute_legacy_launch(l, d_positions, d_values, n, val_dim, subtract_mean, d_distributed, d_indices, d_weights, s);
    if (rc) return rc;
  }
  distribute_remember(l, d_positions, n, val_dim, subtract_mean, d_distributed, d_indices, d_weights);
  l->dist_val = d_values;
  return TLN_OK;
}

// the first halves of up to TLN_BK_MAXJOBS distributes (different lattices, one stream) with FOUR launches for all of
// them; each lattice is finished by its own tln_distribute_finish.  Frames the partitioned kernels do not take
// (val_dim > 1, more than 4M rows, the legacy switch) go through tln_distribute_begin one by one.
extern "C" int tln_distribute_begin_multi(const tln_distribute_call* c, int n, void* stream_) {
  TLN_REQUIRE(c && n >= 1, "bad distribute batch");
  hipStream_t s = (hipStream_t)stream_;
  bool batch = n >= 2 && n <= TLN_BK_MAXJOBS;
  for (int i = 0; i < n; ++i) {
    TLN_REQUIRE(c[i].l && c[i].d_positions && c[i].d_weights, "null argument");
    TLN_REQUIRE(c[i].d_indices || (bk_eligible(c[i].l, c[i].n, c[i].val_dim) && !c[i].d_distributed), "d_indices is required here");
    TLN_REQUIRE(c[i].l->level == 0, "distribute works on the finest level");
    TLN_REQUIRE(c[i].n > 0 && 4 * c[i].n < (1ll << 31), "nr of points %lld out of range", (long long)c[i].n);
    TLN_REQUIRE(c[i].val_dim >= 0 && c[i].val_dim <= 1024 && (c[i].val_dim == 0 || c[i].d_values), "bad val_dim %d", c[i].val_dim);
    for (int k = 0; k < i; ++k) TLN_REQUIRE(c[k].l != c[i].l, "the same lattice twice in one batch");
    if (!bk_eligible(c[i].l, c[i].n, c[i].val_dim)) batch = false;
  }
  if (!batch) {
    for (int i = 0; i < n; ++i) {
      int rc = tln_distribute_begin(c[i].l, c[i].d_positions, c[i].d_values, c[i].n, c[i].val_dim, c[i].subtract_mean,
                                    c[i].d_distributed, c[i].d_indices, c[i].d_weights, stream_);
      if (rc) return rc;
    }
    return TLN_OK;
  }
  BkJobs jobs;
  int split_t = 0;
  for (int i = 0; i < n; ++i) {
    int rc = distribute_prepare(c[i].l, c[i].n, s);
    if (rc) return rc;
  }
  for (int i = 0; i < n; ++i) {
    int st = 0;
    int rc = bk_fill_job(c[i].l, c[i].d_positions, c[i].d_values, c[i].n, c[i].val_dim, c[i].d_distributed, c[i].d_indices,
                         c[i].d_weights, jobs.j[i], &st);
    if (rc) return rc;
    if (st > split_t) split_t = st;
  }
  tln_lattice* l0 = c[0].l;
  if (!l0->ctr_event) TLN_HIP(hipEventCreateWithFlags(&l0->ctr_event, hipEventDisableTiming));
  int rc = bk_launch(jobs, n, split_t, l0->ctr_event, s);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    // ONE event for the batch: the other lattices wait on the first one's (they are finished together; a later
    // re-record by the first lattice only makes a straggler wait for more than it needs)
    c[i].l->ctr_wait = i ? l0->ctr_event : nullptr;
    distribute_remember(c[i].l, c[i].d_positions, c[i].n, c[i].val_dim, c[i].subtract_mean, c[i].d_distributed,
                        c[i].d_indices, c[i].d_weights);
    c[i].l->dist_val = c[i].d_values;
  }
  return TLN_OK;
}

// second half: waits for the counters only (not for work enqueued after them); the mean subtraction of the [4N, .]
// rows when the caller asked for them
extern "C" int tln_distribute_finish(tln_lattice_t* l, void* stream_) {
  TLN_REQUIRE(l && l->dist_pending, "tln_distribute_finish without tln_distribute_begin");
  hipStream_t s = (hipStream_t)stream_;
  l->dist_pending = false;
  TLN_HIP(hipEventSynchronize(l->ctr_wait ? l->ctr_wait : l->ctr_event));
  l->ctr_wait = nullptr;
  if (l->h_ctr[CTR_BUCKET_FULL] != 0) {
    // a bucket of the partitioned kernels met more distinct keys than its LDS table holds (skewed hashing; never on a
    // LiDAR frame at the default geometry): nothing was numbered, the keys entered so far sit un-numbered in the table with
    // their first-touch rows — the per-row-atomic kernels redo the frame from there, with the same result
    if (!l->ctr_event) TLN_HIP(hipEventCreateWithFlags(&l->ctr_event, hipEventDisableTiming));
    TLN_HIP(hipMemsetAsync(l->d_ctr + CTR_BUCKET_FULL, 0, sizeof(int32_t), s));
    int32_t* idx = const_cast<int32_t*>(l->dist_idx);
    int rc = distribute_legacy_launch(l, l->dist_pos, l->dist_val, l->dist_rows / 4, l->dist_val_dim, l->dist_subtract,
                                      l->dist_out, idx, l->dist_w, s);
    if (rc) return rc;
    TLN_HIP(hipEventSynchronize(l->ctr_event));
    ++l->bucket_fallbacks;
  }
  publish_counts(l);
  if (l->h_ctr[CTR_PROBE_FAIL] != 0) {
    tln_set_error("hash probing failed for %d rows (table too full)", l->h_ctr[CTR_PROBE_FAIL]);
    return TLN_E_CAPACITY;
  }
  l->overflow_stale = true;   // counted by k_bins_scatter, after the fetch: read on demand (tln_lattice_overflow_rows)
  const int64_t rows = l->dist_rows;
  l->bins_rows = rows;
  l->bins_dist = l->dist_out;
  l->bins_subtract = l->dist_subtract;
  if (l->dist_out && l->dist_subtract && l->nr_vertices > 0) {
    hipLaunchKernelGGL(k_subtract_rows, dim3((unsigned)tln_cdiv(rows, 256)), dim3(256), 0, s, l->dist_pos, l->dist_idx,
                       l->mean, rows, 3 + l->dist_val_dim + 1, l->dist_out);
    TLN_LAUNCH_CHECK();
  }
  return TLN_OK;
}

extern "C" int tln_distribute(tln_lattice_t* l, const float* d_positions, const float* d_values, int64_t n,
                              int val_dim, int subtract_mean, float* d_distributed, int32_t* d_indices,
                              float* d_weights, void* stream_) {
  int rc = tln_distribute_begin(l, d_positions, d_values, n, val_dim, subtract_mean, d_distributed, d_indices, d_weights,
                                stream_);
  return rc ? rc : tln_distribute_finish(l, stream_);
}

extern "C" int tln_lattice_insert_keys(tln_lattice_t* l, const int32_t* d_keys, int64_t n, int32_t* d_indices_out,
                                       void* stream_) {
  TLN_REQUIRE(l && d_keys && n > 0 && n < (1ll << 31), "bad insert_keys arguments");
  hipStream_t s = (hipStream_t)stream_;
  int rc = ensure_rows(l, n);
  if (rc) return rc;
  rc = ensure_slots(l, n, s);
  if (rc) return rc;
  TableRef t = table_ref(l);
  hipLaunchKernelGGL(k_insert_keys, dim3((unsigned)tln_cdiv(n, 256)), dim3(256), 0, s, d_keys, n, t, l->row_slot);
  TLN_LAUNCH_CHECK();
  rc = number_new(l, n, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_row_indices, dim3((unsigned)tln_cdiv(n, 256)), dim3(256), 0, s, t, l->row_slot, n,
                     d_indices_out, l->d_ctr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  TLN_LAUNCH_CHECK();
  l->csr_rows = -1;
  return fetch_counters(l, s);
}

__global__ void __launch_bounds__(256) k_copy_keys(const int32_t* __restrict__ vkeys, int64_t nv,
                                                   int32_t* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  out[3 * v] = vkeys[4 * v];
  out[3 * v + 1] = vkeys[4 * v + 1];
  out[3 * v + 2] = vkeys[4 * v + 2];
}

extern "C" int tln_lattice_keys(const tln_lattice_t* l, int32_t* d_keys_out, int64_t max_rows, void* stream_) {
  TLN_REQUIRE(l && d_keys_out, "null argument");
  const int64_t nv = l->nr_vertices < max_rows ? l->nr_vertices : max_rows;
  if (nv <= 0) return TLN_OK;
  hipLaunchKernelGGL(k_copy_keys, dim3((unsigned)tln_cdiv(nv, 256)), dim3(256), 0, (hipStream_t)stream_, l->vkeys, nv,
                     d_keys_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

// ---------------------------------------------------------------------------------------
// neighbour tables.  tap order (oracle/permuto.py S4): k=2a -> key+off_a, k=2a+1 -> key-off_a,
// off_a = (1,1,1,1) with -3 at axis a; centre LAST (reference lattice_modules.py:320).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void tap_key(int k0, int k1, int k2, int tap, int& n0, int& n1, int& n2) {
  if (tap == 8) {
    n0 = k0; n1 = k1; n2 = k2;
    return;
  }
  const int a = tap >> 1;
  const int sgn = (tap & 1) ? -1 : 1;
  n0 = k0 + sgn * (a == 0 ? -3 : 1);
  n1 = k1 + sgn * (a == 1 ? -3 : 1);
  n2 = k2 + sgn * (a == 2 ? -3 : 1);
}

// mode 0: same level (query = own keys); mode 1: coarse->fine (query = 2*key into `t` = fine table);
// mode 2: fine->coarse (query = finefy centre of the fine key into `t` = coarse table)
__device__ __forceinline__ void table_entry(const int32_t* __restrict__ qkeys, int64_t nq, const TableRef& t, int mode,
                                            int32_t* __restrict__ table, int64_t gid) {
  const int64_t v = gid / TLN_TAPS;
  const int tap = (int)(gid - v * TLN_TAPS);
  if (v >= nq) return;
  int k0 = qkeys[4 * v], k1 = qkeys[4 * v + 1], k2 = qkeys[4 * v + 2];
  if (mode == 1) {
    k0 *= 2; k1 *= 2; k2 *= 2;
  } else if (mode == 2) {
    int rem0[4], rank[4], bn[4];
    coarse_simplex(k0, k1, k2, rem0, rank, bn);
    int best = 0;
#pragma unroll
    for (int r = 1; r < 4; ++r)
      if (bn[r] > bn[best]) best = r;
    vertex_key(rem0, rank, best, k0, k1, k2);
  }
  int n0, n1, n2;
  tap_key(k0, k1, k2, tap, n0, n1, n2);
  int res = -1;
  if (mode == 0 && tap == 8) res = (int)v;
  else if (tln_key_in_range(n0, n1, n2)) res = probe_find(t, tln_pack_key(n0, n1, n2));
  table[gid] = res;
}

__global__ void __launch_bounds__(256) k_neighbour_table(const int32_t* __restrict__ qkeys, int64_t nq, TableRef t,
                                                         int mode, int32_t* __restrict__ table) {
  table_entry(qkeys, nq, t, mode, table, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// every stale table of a level stack in ONE launch (a frame needs up to 3 + 2 + 2 of them)
struct TableJob {
  const int32_t* qkeys;
  int64_t nq;
  TableRef t;
  int32_t* out;
  int mode;
  int block_begin;
};
#define TLN_TABLE_MAXJOBS 48   // 8 lock-stepped sequences x (3 tables on each of 2 coarse levels)
struct TableJobs {
  TableJob j[TLN_TABLE_MAXJOBS];
  int n;
};
__global__ void __launch_bounds__(256) k_tables_multi(const TableJobs jobs) {
  int k = 0;
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.j[i].block_begin) k = i;
  const TableJob& jb = jobs.j[k];
  table_entry(jb.qkeys, jb.nq, jb.t, jb.mode, jb.out, (int64_t)(blockIdx.x - jb.block_begin) * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------------------
// Row orders of the tap tables.  About 30 % of the (vertex, neighbour) pairs of a LiDAR lattice do not exist (a surface
// in space: 2-7 of the 8 neighbours), and the gather-GEMM multiplies a zero row for each of them.  The large-M kernel
// works on blocks of 128 rows x one tap at a time: if the rows of a block all lack a tap, the whole K chunk of that
// tap can be skipped — which never happens in vertex order (first-touch order of shuffled points) and happens for 23 %
// of the (block, tap) pairs of a level-0 lattice once the rows are ordered by their set of present taps (41 % for the
// coarsen tables).  The output rows of a product are independent, so the kernel may walk them in any order: this is
// that order, one stable 8-bit counting sort per table (key = presence bits of the eight neighbour taps), rebuilt with
// the table.  Deterministic: a wave walks its 256 rows in order, lanes with equal keys are ranked by ballots.
//   k_perm_count    block = 1024 rows: counts per key -> hist[chunk][256]
//   k_perm_scatter  block = 1024 rows: first position of (chunk, key) from the histograms, rows written in order
// gemm.hip asks tln_table_perm(table, M) for every product with a tap table; tables of more than 128k rows have none.
// ---------------------------------------------------------------------------------------
#define TLN_PERM_CHUNK 1024
#define TLN_PERM_MAX_ROWS (1 << 17)
struct PermJob {
  const int32_t* table;
  int64_t rows;
  int32_t* hist;
  int32_t* perm;
  int32_t* order;   // launch order of the 128-row blocks of `perm` (k_tile_order)
};
struct PermJobs {
  PermJob j[TLN_TABLE_MAXJOBS];
  int n;
};
// sort key of a row: the position of its tap set (presence bits of the eight neighbour taps) in the Gray sequence —
// equal sets stay together and neighbouring sets differ in one tap, so a block that straddles a few sets has a small
// union: 75.8 % of the (block, tap) chunks remain on the level-0 table of the headline lattice against 77.4 % for the
// plain value of the bits (82 / 84 % on level 1, 56 / 60 % for coarsen).  Ranking the sets by their number of taps,
// most first, so that the heavy blocks of a launch start first, is worse on both counts (80 % of the chunks remain,
// and 1098 against 1132 clouds/s measured).
__device__ __forceinline__ int perm_key(const int32_t* __restrict__ table, int64_t row) {
  const int32_t* t = table + row * TLN_TAPS;
  int m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) m |= (t[k] >= 0 ? 1 : 0) << k;
  m ^= m >> 1;   // inverse Gray code of the 8 bits
  m ^= m >> 2;
  m ^= m >> 4;
  return m;
}
__global__ void __launch_bounds__(256) k_perm_count(const PermJobs jobs) {
  const PermJob& jb = jobs.j[blockIdx.y];
  const int64_t r0 = (int64_t)blockIdx.x * TLN_PERM_CHUNK;
  if (r0 >= jb.rows) return;
  __shared__ int cnt[256];
  const int tid = threadIdx.x;
  cnt[tid] = 0;
  __syncthreads();
  for (int k = 0; k < TLN_PERM_CHUNK / 256; ++k) {
    const int64_t row = r0 + k * 256 + tid;
    if (row < jb.rows) atomicAdd(&cnt[perm_key(jb.table, row)], 1);
  }
  __syncthreads();
  jb.hist[(int64_t)blockIdx.x * 256 + tid] = cnt[tid];
}
// block = 1024 rows, four waves of 256 consecutive rows each (four steps of 64): thread c first finds where key c of
// this chunk starts (rows of smaller keys anywhere + rows of key c in earlier chunks), the waves' shares of the chunk
// follow from their own counts, then every wave places its rows in order
__global__ void __launch_bounds__(256) k_perm_scatter(const PermJobs jobs) {
  const PermJob& jb = jobs.j[blockIdx.y];
  const int chunk = blockIdx.x;
  const int64_t r0 = (int64_t)chunk * TLN_PERM_CHUNK;
  if (r0 >= jb.rows) return;
  __shared__ int cntw[4][256];   // rows of key c in wave w of this chunk, then the wave's next position for key c
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int k = 0; k < 4; ++k) cntw[k][tid] = 0;
  __syncthreads();
  int keys[4];
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int64_t row = r0 + w * 256 + step * 64 + lane;
    keys[step] = row < jb.rows ? perm_key(jb.table, row) : -1;
    if (keys[step] >= 0) atomicAdd(&cntw[w][keys[step]], 1);
  }
  // key c = tid: rows of this key in all chunks / in the chunks before this one
  const int nchunks = (int)((jb.rows + TLN_PERM_CHUNK - 1) / TLN_PERM_CHUNK);
  int tot = 0, before = 0;
  for (int j = 0; j < nchunks; ++j) {
    const int h = jb.hist[(int64_t)j * 256 + tid];
    tot += h;
    if (j < chunk) before += h;
  }
  int incl = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();   // (also: every wave's counts are complete)
  int pre = 0;
  for (int k = 0; k < w; ++k) pre += wsum[k];
  int pos = pre + incl - tot + before;   // first position of (this chunk, key tid)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cntw[k][tid];
    cntw[k][tid] = pos;                  // wave k's first position for key tid
    pos += c;
  }
  __syncthreads();
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const int64_t row = r0 + w * 256 + step * 64 + lane;
    const int key = keys[step];
    const bool valid = key >= 0;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (key >> bit) & 1;
      const unsigned long long m = __ballot(one);
      peers &= one ? m : ~m;
    }
    // lanes of one key: the lowest one reads and advances the wave's position for the key, the others take theirs from it
    const int leader = valid ? __builtin_ctzll(peers) : lane;
    int base = 0;
    if (valid && leader == lane) {
      base = cntw[w][key];
      cntw[w][key] = base + __popcll(peers);
    }
    base = __shfl(base, leader, 64);
    if (valid) jb.perm[base + __popcll(peers & ((1ull << lane) - 1ull))] = (int32_t)row;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next step reads the advanced positions
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// The launch order of a product's 128-row blocks: heaviest first.  A block of the large-M gather-GEMM multiplies only the
// taps at least one of its rows has (3 .. 9 of 9 on the level-0 table of the headline lattice, mean 6.8), and blocks are
// dispatched in index order — in the row order's own sequence the heavy blocks of the last sequence of a shared launch
// start last and the launch ends on them (simulated on that lattice: 9.1 % over the balanced time, against 2.6 % with
// every sequence's blocks heaviest first).  Two small launches per batch of tables:
//   k_tile_cost   block = one 128-row block of the row order: union of its rows' tap sets -> number of taps
//   k_tile_order  one workgroup per table: stable descending counting rank of the <= 1024 blocks -> order[rank] = block
#define TLN_TILE_ROWS 128
__global__ void __launch_bounds__(TLN_TILE_ROWS) k_tile_cost(const PermJobs jobs) {
  const PermJob& jb = jobs.j[blockIdx.y];
  const int64_t p = (int64_t)blockIdx.x * TLN_TILE_ROWS + threadIdx.x;
  if ((int64_t)blockIdx.x * TLN_TILE_ROWS >= jb.rows) return;
  __shared__ unsigned part[TLN_TILE_ROWS / 64];
  unsigned mine = 0;
  if (p < jb.rows) {
    const int32_t* t = jb.table + (int64_t)jb.perm[p] * TLN_TAPS;
#pragma unroll
    for (int k = 0; k < 8; ++k) mine |= (t[k] >= 0 ? 1u : 0u) << k;
  }
  unsigned u = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (__ballot((mine >> k) & 1u) != 0ull) u |= 1u << k;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = u;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < TLN_TILE_ROWS / 64; ++w) u |= part[w];
    jb.hist[blockIdx.x] = __popc(u);     // (the histogram scratch of the row order is free again)
  }
}
__global__ void __launch_bounds__(1024) k_tile_order(const PermJobs jobs) {
  const PermJob& jb = jobs.j[blockIdx.x];
  const int n = (int)((jb.rows + TLN_TILE_ROWS - 1) / TLN_TILE_ROWS);
  __shared__ int cost[1024];
  const int t = threadIdx.x;
  cost[t] = t < n ? jb.hist[t] : -1;
  __syncthreads();
  if (t >= n) return;
  const int c = cost[t];
  int rank = 0;
  for (int j = 0; j < n; ++j) {
    const int cj = cost[j];
    rank += (cj > c || (cj == c && j < t)) ? 1 : 0;
  }
  jb.order[rank] = t;
}

// table -> its row order, for gemm.hip (several lattices on several host threads: a lock around a small map)
#include <mutex>
#include <unordered_map>
struct PermInfo {
  const int32_t* perm;
  int64_t rows;
  const int32_t* order;
};
static std::mutex g_perm_mu;
static std::unordered_map<const int32_t*, PermInfo> g_perm;
static bool perm_enabled() {
  static const bool off = getenv("TLN_PERM_OFF") != nullptr;
  return !off;
}
const int32_t* tln_table_perm(const int32_t* table, int64_t rows) {
  if (!table || !perm_enabled()) return nullptr;
  std::lock_guard<std::mutex> lk(g_perm_mu);
  auto it = g_perm.find(table);
  return (it != g_perm.end() && it->second.rows == rows) ? it->second.perm : nullptr;
}
static bool tile_order_enabled() {
  static const bool off = getenv("TLN_TILE_ORDER_OFF") != nullptr;
  return !off;
}
// the launch order of the table's 128-row blocks (heaviest first), if its row order was built for exactly `rows` rows
const int32_t* tln_table_tile_order(const int32_t* table, int64_t rows) {
  if (!table || !perm_enabled() || !tile_order_enabled()) return nullptr;
  std::lock_guard<std::mutex> lk(g_perm_mu);
  auto it = g_perm.find(table);
  return (it != g_perm.end() && it->second.rows == rows) ? it->second.order : nullptr;
}
static void perm_forget(const int32_t* table) {
  if (!table) return;
  std::lock_guard<std::mutex> lk(g_perm_mu);
  g_perm.erase(table);
}
// the row orders of up to TLN_TABLE_MAXJOBS freshly built tables: two launches for all of them
struct PermWant {
  const int32_t* table;
  int64_t rows;
  int32_t** perm;     // the level's buffers (allocated here on first use)
  int32_t** hist;
  int64_t capacity;   // rows the table can hold
};
static int build_perms(const PermWant* w, int n, hipStream_t s) {
  if (!perm_enabled()) return TLN_OK;
  PermJobs jobs{};
  int64_t maxchunks = 0;
  for (int i = 0; i < n; ++i) {
    if (w[i].rows <= 0 || w[i].rows > TLN_PERM_MAX_ROWS) {
      perm_forget(w[i].table);
      continue;
    }
    if (!*w[i].perm) {
      const int64_t cap = w[i].capacity < TLN_PERM_MAX_ROWS ? w[i].capacity : TLN_PERM_MAX_ROWS;
      TLN_HIP(hipMalloc(w[i].perm, (size_t)(cap + cap / TLN_TILE_ROWS + 8) * sizeof(int32_t)));   // rows, then the block order
      TLN_HIP(hipMalloc(w[i].hist, (size_t)(cap / TLN_PERM_CHUNK + 2) * 256 * sizeof(int32_t)));
    }
    PermJob& jb = jobs.j[jobs.n++];
    jb.table = w[i].table;
    jb.rows = w[i].rows;
    jb.hist = *w[i].hist;
    jb.perm = *w[i].perm;
    {
      const int64_t cap = w[i].capacity < TLN_PERM_MAX_ROWS ? w[i].capacity : TLN_PERM_MAX_ROWS;
      jb.order = *w[i].perm + cap;
    }
    const int64_t ch = tln_cdiv(w[i].rows, TLN_PERM_CHUNK);
    if (ch > maxchunks) maxchunks = ch;
    std::lock_guard<std::mutex> lk(g_perm_mu);
    g_perm[w[i].table] = PermInfo{*w[i].perm, w[i].rows, jb.order};
  }
  if (jobs.n == 0) return TLN_OK;
  hipLaunchKernelGGL(k_perm_count, dim3((unsigned)maxchunks, (unsigned)jobs.n), dim3(256), 0, s, jobs);
  hipLaunchKernelGGL(k_perm_scatter, dim3((unsigned)maxchunks, (unsigned)jobs.n), dim3(256), 0, s, jobs);
  if (tile_order_enabled()) {
    hipLaunchKernelGGL(k_tile_cost, dim3((unsigned)(maxchunks * (TLN_PERM_CHUNK / TLN_TILE_ROWS)), (unsigned)jobs.n), dim3(TLN_TILE_ROWS), 0, s, jobs);
    hipLaunchKernelGGL(k_tile_order, dim3((unsigned)jobs.n), dim3(1024), 0, s, jobs);
  }
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

static int ensure_table(int32_t** p, int64_t capacity, int64_t* cap_out) {
  if (*p) return TLN_OK;
  TLN_HIP(hipMalloc(p, capacity * TLN_TAPS * sizeof(int32_t)));
  *cap_out = capacity;
  return TLN_OK;
}

extern "C" int tln_neighbour_table(tln_lattice_t* l, const int32_t** d_table_out, void* stream_) {
  TLN_REQUIRE(l && d_table_out, "null argument");
  int rc = ensure_table(&l->nbr, l->capacity, &l->table_cap_nbr);
  if (rc) return rc;
  if (l->nbr_gen != l->gen && l->nr_vertices > 0) {
    const int64_t total = l->nr_vertices * TLN_TAPS;
    hipLaunchKernelGGL(k_neighbour_table, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_,
                       l->vkeys, l->nr_vertices, table_ref(l), 0, l->nbr);
    TLN_LAUNCH_CHECK();
    l->nbr_gen = l->gen;
    const PermWant w{l->nbr, l->nr_vertices, &l->perm_nbr, &l->phist_nbr, l->capacity};
    rc = build_perms(&w, 1, (hipStream_t)stream_);
    if (rc) return rc;
  }
  *d_table_out = l->nbr;
  return TLN_OK;
}

static int finish_pending(tln_lattice* l, void* stream_) {
  tln_lattice* root = l;
  while (root->parent) root = root->parent;
  return root->levels_pending ? tln_lattice_prepare_levels_finish(root, stream_) : TLN_OK;
}

extern "C" int tln_coarsen(tln_lattice_t* fine, tln_lattice_t** coarse_out, void* stream_) {
  TLN_REQUIRE(fine && coarse_out, "null argument");
  {
    int rc = finish_pending(fine, stream_);
    if (rc) return rc;
  }
  hipStream_t s = (hipStream_t)stream_;
  if (!fine->coarse) {
    double sg[3] = {fine->sigmas[0] * 2, fine->sigmas[1] * 2, fine->sigmas[2] * 2};
    tln_lattice* c = nullptr;
    int rc = lattice_alloc(&c, fine->pos_dim, sg, fine->capacity, fine->level + 1, fine->scale_constant);
    if (rc) return rc;
    c->parent = fine;
    fine->coarse = c;
    rc = tln_lattice_clear(c, s);
    if (rc) return rc;
  }
  tln_lattice* c = fine->coarse;
  const int64_t first = c->embedded_fine, count = fine->nr_vertices - first;
  if (count > 0) {
    const int64_t rows = 4 * count;
    int rc = ensure_rows(c, rows);
    if (rc) return rc;
    rc = ensure_slots(c, rows, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_coarsen_insert, dim3((unsigned)tln_cdiv(count, 256)), dim3(256), 0, s, fine->vkeys, first,
                       count, table_ref(c), c->row_slot, (const int32_t*)nullptr);
    TLN_LAUNCH_CHECK();
    rc = number_new(c, rows, s);
    if (rc) return rc;
    rc = fetch_counters(c, s);
    if (rc) return rc;
    c->embedded_fine = fine->nr_vertices;
    c->csr_rows = -1;
  }
  *coarse_out = c;
  return TLN_OK;
}

// tln_coarsen without the host round trip: `fine_bound` >= the fine level's current vertex count (exact for level 0,
// previous count + 4 x the finer level's growth otherwise); the kernels read the true count from the fine level's
// device counters.  The caller fetches the new counts of all levels at once afterwards.
static int coarsen_deferred(tln_lattice* fine, int64_t fine_bound, hipStream_t s) {
  if (!fine->coarse) {
    double sg[3] = {fine->sigmas[0] * 2, fine->sigmas[1] * 2, fine->sigmas[2] * 2};
    tln_lattice* c = nullptr;
    int rc = lattice_alloc(&c, fine->pos_dim, sg, fine->capacity, fine->level + 1, fine->scale_constant);
    if (rc) return rc;
    c->parent = fine;
    fine->coarse = c;
    rc = tln_lattice_clear(c, s);
    if (rc) return rc;
  }
  tln_lattice* c = fine->coarse;
  if (fine_bound > fine->capacity) fine_bound = fine->capacity;
  const int64_t first = c->embedded_fine, bound = fine_bound - first;
  if (bound <= 0) return TLN_OK;
  const int64_t rows = 4 * bound;
  int rc = ensure_rows(c, rows);
  if (rc) return rc;
  rc = ensure_slots(c, rows, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_coarsen_insert, dim3((unsigned)tln_cdiv(bound, 256)), dim3(256), 0, s, fine->vkeys, first, bound,
                     table_ref(c), c->row_slot, (const int32_t*)fine->d_ctr);
  TLN_LAUNCH_CHECK();
  rc = number_new(c, rows, s);
  if (rc) return rc;
  c->csr_rows = -1;
  return TLN_OK;
}

extern "C" int tln_coarse_to_fine_table(tln_lattice_t* c, const int32_t** d_table_out, void* stream_) {
  TLN_REQUIRE(c && c->parent && d_table_out, "not a coarse level");
  {
    int rc = finish_pending(c, stream_);
    if (rc) return rc;
  }
  tln_lattice* f = c->parent;
  int rc = ensure_table(&c->c2f, c->capacity, &c->table_cap_c2f);
  if (rc) return rc;
  if ((c->c2f_gen_c != c->gen || c->c2f_gen_f != f->gen) && c->nr_vertices > 0) {
    const int64_t total = c->nr_vertices * TLN_TAPS;
    hipLaunchKernelGGL(k_neighbour_table, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_,
                       c->vkeys, c->nr_vertices, table_ref(f), 1, c->c2f);
    TLN_LAUNCH_CHECK();
    c->c2f_gen_c = c->gen;
    c->c2f_gen_f = f->gen;
    const PermWant w{c->c2f, c->nr_vertices, &c->perm_c2f, &c->phist_c2f, c->capacity};
    rc = build_perms(&w, 1, (hipStream_t)stream_);
    if (rc) return rc;
  }
  *d_table_out = c->c2f;
  return TLN_OK;
}

extern "C" int tln_fine_to_coarse_table(tln_lattice_t* c, const int32_t** d_table_out, void* stream_) {
  TLN_REQUIRE(c && c->parent && d_table_out, "not a coarse level");
  {
    int rc = finish_pending(c, stream_);
    if (rc) return rc;
  }
  tln_lattice* f = c->parent;
  int rc = ensure_table(&c->f2c, f->capacity, &c->table_cap_f2c);
  if (rc) return rc;
  if ((c->f2c_gen_c != c->gen || c->f2c_gen_f != f->gen) && f->nr_vertices > 0) {
    const int64_t total = f->nr_vertices * TLN_TAPS;
    hipLaunchKernelGGL(k_neighbour_table, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream_,
                       f->vkeys, f->nr_vertices, table_ref(c), 2, c->f2c);
    TLN_LAUNCH_CHECK();
    c->f2c_gen_c = c->gen;
    c->f2c_gen_f = f->gen;
    const PermWant w{c->f2c, f->nr_vertices, &c->perm_f2c, &c->phist_f2c, f->capacity};
    rc = build_perms(&w, 1, (hipStream_t)stream_);
    if (rc) return rc;
  }
  *d_table_out = c->f2c;
  return TLN_OK;
}

// Build the coarse levels and every stale table of the stack in as few launches as possible: per coarse level one
// insertion + one numbering launch, then ONE launch for all neighbour / cross-level tables.  Called once per frame right
// after tln_distribute; the per-table getters then only return pointers.
// prepare_levels in two halves so that a caller can launch work that only needs level 0 while the coarse levels'
// vertex counts are still on their way to the host:
//   _begin   extends every coarse level (no host round trip in between; the numbering kernels write the counters into
//            the host's mapped words), records an event, builds the level-0 neighbour table; v_bound_out[0] = V0
//            (exact), v_bound_out[i] >= the new vertex count of level i
//   _finish  waits for that event (not for the stream), publishes the exact counts, builds the coarse tables
// Both halves take the level stacks of up to TLN_LVL_MAXJOBS lattices (the lock-stepped sequences of a stream): every
// launch then carries the work of all of them (blockIdx.y / job list = lattice).

// the tables of `jobs` (built into them by add_table) + their row orders, TLN_TABLE_MAXJOBS per launch
struct TableBatch {
  TableJobs jobs{};
  PermWant pw[TLN_TABLE_MAXJOBS];
  int blocks = 0;
};
static int flush_tables(TableBatch& tb, hipStream_t s) {
  if (tb.jobs.n == 0) return TLN_OK;
  hipLaunchKernelGGL(k_tables_multi, dim3((unsigned)tb.blocks), dim3(256), 0, s, tb.jobs);
  TLN_LAUNCH_CHECK();
  int rc = build_perms(tb.pw, tb.jobs.n, s);
  tb.jobs.n = 0;
  tb.blocks = 0;
  return rc;
}
static int add_table(TableBatch& tb, const int32_t* qkeys, int64_t nq, tln_lattice* target, int mode, int32_t* out,
                     const PermWant& w, hipStream_t s) {
  if (tb.jobs.n == TLN_TABLE_MAXJOBS) {
    int rc = flush_tables(tb, s);
    if (rc) return rc;
  }
  tb.pw[tb.jobs.n] = w;
  TableJob& j = tb.jobs.j[tb.jobs.n++];
  j.qkeys = qkeys;
  j.nq = nq;
  j.t = table_ref(target);
  j.mode = mode;
  j.out = out;
  j.block_begin = tb.blocks;
  tb.blocks += (int)tln_cdiv(nq * TLN_TAPS, 256);
  return TLN_OK;
}

static int prepare_levels_begin_multi(tln_lattice* const* ll, int n, int nr_coarse_levels, int64_t* v_bound_out, int vstride,
                                      hipStream_t s) {
  TLN_REQUIRE(ll && n >= 1 && n <= TLN_LVL_MAXJOBS && nr_coarse_levels >= 0 && nr_coarse_levels <= 3,
              "bad prepare_levels arguments");
  tln_lattice* lv[TLN_LVL_MAXJOBS];
  int64_t bound[TLN_LVL_MAXJOBS], growth[TLN_LVL_MAXJOBS];
  for (int i = 0; i < n; ++i) {
    tln_lattice* l0 = ll[i];
    TLN_REQUIRE(l0 && !l0->parent, "prepare_levels wants the finest level");
    if (l0->levels_pending) {
      int rc = tln_lattice_prepare_levels_finish(l0, s);
      if (rc) return rc;
    }
    if (v_bound_out) v_bound_out[(size_t)i * vstride] = l0->nr_vertices;
    lv[i] = l0;
    bound[i] = l0->nr_vertices;                       // exact
    growth[i] = l0->nr_vertices - (l0->coarse ? l0->coarse->embedded_fine : 0);
  }
  for (int lev = 0; lev < nr_coarse_levels; ++lev) {
    LevelJobs jobs;
    int64_t max_bound = 0;
    int max_blocks = 0, m = 0;
    bool any_small = false, any_large = false;
    for (int i = 0; i < n; ++i) {
      tln_lattice* fine = lv[i];
      if (!fine->coarse) {
        double sg[3] = {fine->sigmas[0] * 2, fine->sigmas[1] * 2, fine->sigmas[2] * 2};
        tln_lattice* c = nullptr;
        int rc = lattice_alloc(&c, fine->pos_dim, sg, fine->capacity, fine->level + 1, fine->scale_constant);
        if (rc) return rc;
        c->parent = fine;
        fine->coarse = c;
        tln_lattice* cc = c;
        rc = tln_lattice_clear_multi(&cc, 1, s);
        if (rc) return rc;
      }
      tln_lattice* c = fine->coarse;
      int64_t fb = bound[i] > fine->capacity ? fine->capacity : bound[i];
      const int64_t first = c->embedded_fine, b = fb - first;
      if (b > 0) {
        const int64_t rows = 4 * b;
        int rc = ensure_rows(c, rows);
        if (rc) return rc;
        rc = ensure_slots(c, rows, s);
        if (rc) return rc;
        LevelJob& J = jobs.j[m++];
        J.fine_keys = fine->vkeys;
        J.fine_ctr = fine->d_ctr;
        J.row_slot = c->row_slot;
        J.ctr = c->d_ctr;
        J.host_ctr = c->h_ctr_dev;
        J.vkeys = c->vkeys;
        J.vslot = c->vslot;
        J.block_cnt = c->block_cnt;
        J.t = table_ref(c);
        J.first = first;
        J.bound = b;
        J.rows = rows;
        J.capacity = (int)c->capacity;
        J.vold = (int)c->nr_vertices;
        J.nblocks = (int)tln_cdiv(rows, TLN_SCAN_BLOCK);
        J.small = rows <= 16 * TLN_SCAN_BLOCK ? 1 : 0;
        if (J.small) any_small = true;
        else any_large = true;
        if (b > max_bound) max_bound = b;
        if (!J.small && J.nblocks > max_blocks) max_blocks = J.nblocks;
        c->csr_rows = -1;
      }
      if (growth[i] < 0) growth[i] = 0;
      growth[i] *= 4;                                      // a fine vertex touches at most 4 coarse vertices
      bound[i] = c->nr_vertices + growth[i];               // >= the coarse level's new count
      if (bound[i] > c->capacity) bound[i] = c->capacity;
      if (v_bound_out) v_bound_out[(size_t)i * vstride + lev + 1] = bound[i];
      lv[i] = c;
    }
    if (m > 0) {
      for (int k = m; k < TLN_LVL_MAXJOBS; ++k) jobs.j[k] = jobs.j[0];   // (never indexed: the grids have m rows)
      hipLaunchKernelGGL(k_coarsen_insert_m, dim3((unsigned)tln_cdiv(max_bound, 256), (unsigned)m), dim3(256), 0, s, jobs);
      if (any_small) hipLaunchKernelGGL(k_number_small_m, dim3(1, (unsigned)m), dim3(TLN_SCAN_BLOCK), 0, s, jobs);
      if (any_large) {
        hipLaunchKernelGGL(k_count_new_m, dim3((unsigned)max_blocks, (unsigned)m), dim3(TLN_SCAN_BLOCK), 0, s, jobs);
        hipLaunchKernelGGL(k_assign_new_m, dim3((unsigned)max_blocks, (unsigned)m), dim3(TLN_SCAN_BLOCK), 0, s, jobs);
      }
      TLN_LAUNCH_CHECK();
    }
  }
  // ONE event for the batch (the counters of every coarse level are in the hosts' mapped words behind it)
  tln_lattice* first = ll[0];
  if (!first->levels_event) TLN_HIP(hipEventCreateWithFlags(&first->levels_event, hipEventDisableTiming));
  TLN_HIP(hipEventRecord(first->levels_event, s));
  TableBatch tb;
  for (int i = 0; i < n; ++i) {
    tln_lattice* l0 = ll[i];
    l0->levels_wait = i ? first->levels_event : nullptr;
    l0->levels_pending = nr_coarse_levels > 0 ? nr_coarse_levels : -1;   // -1: nothing to wait for, tables only
    if (l0->nr_vertices > 0) {
      int rc = ensure_table(&l0->nbr, l0->capacity, &l0->table_cap_nbr);
      if (rc) return rc;
      if (l0->nbr_gen != l0->gen) {
        rc = add_table(tb, l0->vkeys, l0->nr_vertices, l0, 0, l0->nbr,
                       PermWant{l0->nbr, l0->nr_vertices, &l0->perm_nbr, &l0->phist_nbr, l0->capacity}, s);
        if (rc) return rc;
        l0->nbr_gen = l0->gen;
      }
    }
  }
  return flush_tables(tb, s);
}

static int prepare_levels_finish_multi(tln_lattice* const* ll, int n, hipStream_t s) {
  TableBatch tb;
  for (int i = 0; i < n; ++i) {
    tln_lattice* l0 = ll[i];
    TLN_REQUIRE(l0 && !l0->parent, "prepare_levels wants the finest level");
    if (!l0->levels_pending) continue;
    const int nr_coarse_levels = l0->levels_pending > 0 ? l0->levels_pending : 0;
    l0->levels_pending = 0;
    if (nr_coarse_levels > 0) {
      TLN_HIP(hipEventSynchronize(l0->levels_wait ? l0->levels_wait : l0->levels_event));
      l0->levels_wait = nullptr;
      int lvl = 0;
      for (tln_lattice* c = l0->coarse; c && lvl < nr_coarse_levels; c = c->coarse, ++lvl) {
        set_vertices(c, c->h_ctr[CTR_NV]);
        c->occupied = c->h_ctr[CTR_OCCUPIED];
        c->embedded_fine = c->parent->nr_vertices;
        if (c->h_ctr[CTR_PROBE_FAIL] != 0) {
          tln_set_error("hash probing failed for %d rows at level %d (table too full)", c->h_ctr[CTR_PROBE_FAIL], c->level);
          return TLN_E_CAPACITY;
        }
      }
    }
    for (tln_lattice* p = l0; p; p = p->coarse) {
      if (p->nr_vertices <= 0) continue;
      int rc = ensure_table(&p->nbr, p->capacity, &p->table_cap_nbr);
      if (rc) return rc;
      if (p->nbr_gen != p->gen) {
        rc = add_table(tb, p->vkeys, p->nr_vertices, p, 0, p->nbr,
                       PermWant{p->nbr, p->nr_vertices, &p->perm_nbr, &p->phist_nbr, p->capacity}, s);
        if (rc) return rc;
        p->nbr_gen = p->gen;
      }
      if (p->parent && p->parent->nr_vertices > 0) {
        tln_lattice* f = p->parent;
        rc = ensure_table(&p->c2f, p->capacity, &p->table_cap_c2f);
        if (rc) return rc;
        rc = ensure_table(&p->f2c, f->capacity, &p->table_cap_f2c);
        if (rc) return rc;
        if (p->c2f_gen_c != p->gen || p->c2f_gen_f != f->gen) {
          rc = add_table(tb, p->vkeys, p->nr_vertices, f, 1, p->c2f,
                         PermWant{p->c2f, p->nr_vertices, &p->perm_c2f, &p->phist_c2f, p->capacity}, s);
          if (rc) return rc;
          p->c2f_gen_c = p->gen;
          p->c2f_gen_f = f->gen;
        }
        if (p->f2c_gen_c != p->gen || p->f2c_gen_f != f->gen) {
          rc = add_table(tb, f->vkeys, f->nr_vertices, p, 2, p->f2c,
                         PermWant{p->f2c, f->nr_vertices, &p->perm_f2c, &p->phist_f2c, f->capacity}, s);
          if (rc) return rc;
          p->f2c_gen_c = p->gen;
          p->f2c_gen_f = f->gen;
        }
      }
    }
  }
  return flush_tables(tb, s);
}

extern "C" int tln_lattice_prepare_levels_begin(tln_lattice_t* l0, int nr_coarse_levels, int64_t* v_bound_out,
                                                void* stream_) {
  return prepare_levels_begin_multi(&l0, 1, nr_coarse_levels, v_bound_out, TLN_MAX_LEVELS, (hipStream_t)stream_);
}

extern "C" int tln_lattice_prepare_levels_finish(tln_lattice_t* l0, void* stream_) {
  return prepare_levels_finish_multi(&l0, 1, (hipStream_t)stream_);
}

// the same for the level stacks of n lock-stepped sequences; v_bound_out is [n][TLN_MAX_LEVELS] (or NULL)
extern "C" int tln_lattice_prepare_levels_begin_multi(tln_lattice_t* const* l0, int n, int nr_coarse_levels,
                                                      int64_t* v_bound_out, void* stream_) {
  TLN_REQUIRE(l0 && n >= 1, "null argument");
  for (int i0 = 0; i0 < n; i0 += TLN_LVL_MAXJOBS) {
    const int m = n - i0 < TLN_LVL_MAXJOBS ? n - i0 : TLN_LVL_MAXJOBS;
    int rc = prepare_levels_begin_multi(l0 + i0, m, nr_coarse_levels,
                                        v_bound_out ? v_bound_out + (size_t)i0 * TLN_MAX_LEVELS : nullptr, TLN_MAX_LEVELS,
                                        (hipStream_t)stream_);
    if (rc) return rc;
  }
  return TLN_OK;
}

extern "C" int tln_lattice_prepare_levels_finish_multi(tln_lattice_t* const* l0, int n, void* stream_) {
  TLN_REQUIRE(l0 && n >= 1, "null argument");
  return prepare_levels_finish_multi(l0, n, (hipStream_t)stream_);
}

extern "C" int tln_lattice_prepare_levels(tln_lattice_t* l0, int nr_coarse_levels, void* stream_) {
  int rc = tln_lattice_prepare_levels_begin(l0, nr_coarse_levels, nullptr, stream_);
  if (rc) return rc;
  return tln_lattice_prepare_levels_finish(l0, stream_);
}

// the coarse level of `l` as it stands (NULL if none yet); no side effects
extern "C" tln_lattice_t* tln_lattice_coarse_level(tln_lattice_t* l) { return l ? l->coarse : nullptr; }

// Device memory this handle owns, by purpose, over the whole level stack ("~2 GB per resident sequence,
// unexplained").  Computed from the sizes the allocations below were made with (every array here is sized by the
// level's capacity, slot count or row capacity).  out: [0] hash tables + per-vertex arrays, [1] per-row workspaces of the
// distribute (records, bins, sort scratch), [2] the pool's packed accumulators, [3] neighbour / cross-level tables and
// their row orders, [4] total.
static void lattice_memory_one(const tln_lattice* l, int64_t* out) {
  const int64_t cap = l->capacity, ns = l->nslots, rc = l->rows_cap;
  int64_t a = ns * (int64_t)sizeof(TlnSlot) + cap * 16 + (cap + 2) * 4 + cap * 12 + CTR_COUNT * 4;
  if (l->slot_cnt) a += ns * 4;
  if (l->vslot) a += cap * 16;   // vslot, vcnt, vstart, vstamp
  int64_t b = 0;
  if (rc > 0) {
    b += 5 * rc * 4 + (rc / TLN_SCAN_BLOCK + 2) * 4 + (rc / 256 + 2) * 48 + (int64_t)l->sort_temp_bytes;
    if (l->bin_rec) b += rc * 4 + rc * (int64_t)sizeof(TlnBinRec) + l->rec_cap * 16 + (rc / 4 + 1) * 16 +
                         (int64_t)TLN_BK_SPLIT_BLOCKS * (l->bk_maxb + 1) * 4 + rc * 4 + (int64_t)l->bk_maxb * 4 + (rc / 128 + 8) * 4;
  }
  const int64_t c = l->pool_packed_elems * 8;
  int64_t d = 0;
  auto table = [&](const int32_t* t, int64_t rows_cap_t, const int32_t* perm) {
    if (t) d += rows_cap_t * TLN_TAPS * 4;
    if (perm) {
      const int64_t pc = rows_cap_t < TLN_PERM_MAX_ROWS ? rows_cap_t : TLN_PERM_MAX_ROWS;
      d += (pc + pc / TLN_TILE_ROWS + 8) * 4 + (pc / TLN_PERM_CHUNK + 2) * 256 * 4;
    }
  };
  table(l->nbr, l->table_cap_nbr, l->perm_nbr);
  table(l->c2f, l->table_cap_c2f, l->perm_c2f);
  table(l->f2c, l->table_cap_f2c, l->perm_f2c);
  out[0] += a;
  out[1] += b;
  out[2] += c;
  out[3] += d;
}
extern "C" int tln_lattice_memory(const tln_lattice_t* l, int64_t* out) {
  TLN_REQUIRE(l && out, "null argument");
  for (int i = 0; i < 5; ++i) out[i] = 0;
  for (const tln_lattice* q = l; q; q = q->coarse) lattice_memory_one(q, out);
  out[4] = out[0] + out[1] + out[2] + out[3];
  return TLN_OK;
}
