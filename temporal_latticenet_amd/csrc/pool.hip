// K2 PointNet pool + K11 splat on the vertex-sorted row list (CSR) left by tln_distribute.
//
// Mirrors PointNetSeqModule.forward, reference seq_lattice/lattice_modules.py:448-530:
//   per-row MLP (lm:460-473) -> scatter_max with argmax (lm:512) -> argmax clamp quirk (lm:513-514)
//   -> points-per-vertex (lm:519-521) -> barycentric of the argmax row (lm:522-525) -> <4 mask (lm:527-530)
// without materialising the [4N,64] activations: a wave takes 64 consecutive SORTED rows, every lane runs the
// MLP of its row with wave-uniform weights, the 64x64 result is transposed through LDS and each lane then owns
// one channel and walks the rows once, flushing (max, argmax-row) per vertex segment with one 64-bit atomicMax
// per channel (512 contiguous bytes per wave-instruction).
#include "pool_common.h"

struct tln_lattice;
const int32_t* tln_lat_order(const tln_lattice* l);
const int32_t* tln_lat_sorted_vertex(const tln_lattice* l);
const int32_t* tln_lat_seg_start(const tln_lattice* l);
int64_t tln_lat_csr_rows(const tln_lattice* l);
int tln_lat_pool_ws(tln_lattice* l, int64_t elems, unsigned long long** out);

// CIN -> H1 -> (H2 ->) COUT ; H2 == 0 means a two-layer MLP; H1 == 0 means identity (COUT == CIN).
//
// One wave takes 64 consecutive SORTED rows (rows of one vertex are contiguous).
//   phase 1: lane = row.  The front layers (all but the last) run per lane with wave-uniform weights that the
//            compiler keeps in SGPRs (few hundred scalars); the last hidden activation HL goes to a wave-private
//            LDS tile [64 rows][HL] (rows padded to 16-byte multiples).
//   phase 2: lane = OUTPUT CHANNEL of the last layer.  Its weight row sits in HL registers (loaded once per wave);
//            the rows are walked in order, their activations arrive as LDS broadcast reads (ds_read_b128, one address
//            for the whole wave) and the running (max, argmax-row) of the current vertex lives in registers — no
//            transposition, no per-row scalar-load stalls in the layer that holds 78 % of the arithmetic.  When the
//            vertex changes (wave-uniform branch) the lane flushes its channel with one 64-bit atomicMax of
//            (order-preserving value bits, ~row): 512 contiguous bytes per wave-instruction.
template <int CIN, int H1, int H2, int COUT>
__global__ void __launch_bounds__(256) k_pool_chunks(const float* __restrict__ dist, int cols,
                                                     const int32_t* __restrict__ order,
                                                     const int32_t* __restrict__ sorted_vertex, int64_t rows, int nv,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     const float* __restrict__ w3, const float* __restrict__ b3,
                                                     unsigned long long* __restrict__ packed) {
  constexpr int HL = (H1 == 0) ? CIN : (H2 ? H2 : H1);   // width of the activation that feeds the last layer
  constexpr int TS = ((HL + 3) / 4) * 4 + 4;             // tile row stride in floats (16-byte aligned rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tile = smem + wid * (64 * TS + 128);
  int* tv = reinterpret_cast<int*>(tile + 64 * TS);
  int* tr = tv + 64;

  const int64_t chunk = (int64_t)blockIdx.x * 4 + wid;
  const int64_t j0 = chunk * 64;
  if (j0 >= rows) return;  // no block barrier below: the tile is private to the wave
  const int cnt = (int)((rows - j0) < 64 ? (rows - j0) : 64);

  // ---- phase 1
  if (lane < cnt) {
    const int row = order[j0 + lane];
    int v = sorted_vertex[j0 + lane];
    if (v >= nv) v = 0;  // rows without a vertex fold into vertex 0 (lm:480)
    float x[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) x[c] = dist[(int64_t)row * cols + c];
    float hl[HL];
    if constexpr (H1 == 0) {
#pragma unroll
      for (int c = 0; c < HL; ++c) hl[c] = x[c];
    } else if constexpr (H2 == 0) {
      dense<CIN, H1, true>(w1, b1, x, hl);
    } else {
      float h1[H1];
      dense<CIN, H1, true>(w1, b1, x, h1);
      dense<H1, H2, true>(w2, b2, h1, hl);
    }
#pragma unroll
    for (int c = 0; c < HL; ++c) tile[lane * TS + c] = hl[c];
    tv[lane] = v;
    tr[lane] = row;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2
  const int c = lane;
  const bool active = c < COUT;
  float wl[HL];
  float bias = 0.0f;
  if constexpr (H1 != 0) {
    const float* wlast = H2 ? w3 : w2;
    const float* blast = H2 ? b3 : b2;
#pragma unroll
    for (int i = 0; i < HL; ++i) wl[i] = active ? wlast[c * HL + i] : 0.0f;
    bias = active ? blast[c] : 0.0f;
  }
  auto value_of = [&](int j) {
    const float* h = tile + j * TS;
    if constexpr (H1 == 0) {
      return h[active ? c : 0];
    } else {
      // ONE fma chain in ascending i, as in the front layers: the summation order is part of the specification
      // (oracle/csrc/pool_mlp.c) because the arg-max row decides a whole output element (lm:513-525)
      float acc = bias;
#pragma unroll
      for (int i = 0; i < HL; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(h + i);   // one address for the whole wave: broadcast
        acc = fmaf(wl[i], q.x, acc);
        acc = fmaf(wl[i + 1], q.y, acc);
        acc = fmaf(wl[i + 2], q.z, acc);
        acc = fmaf(wl[i + 3], q.w, acc);
      }
      return acc;
    }
  };
  int cur = tv[0];
  float best = value_of(0);
  int brow = tr[0];
  for (int j = 1; j < cnt; ++j) {
    const int v = tv[j];
    const float val = value_of(j);
    if (v != cur) {  // wave-uniform
      if (active) {
        const unsigned long long p = ((unsigned long long)tln_f2ord(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)brow);
        atomicMax(&packed[(int64_t)cur * COUT + c], p);
      }
      cur = v;
      best = val;
      brow = tr[j];
    } else if (val > best) {  // rows ascend inside a segment: strict '>' keeps the smallest row on ties
      best = val;
      brow = tr[j];
    }
  }
  if (active) {
    const unsigned long long p = ((unsigned long long)tln_f2ord(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)brow);
    atomicMax(&packed[(int64_t)cur * COUT + c], p);
  }
}

__global__ void __launch_bounds__(256) k_pool_finalize(const unsigned long long* packed,
                                                       const int32_t* __restrict__ seg_start, int nv, int cout,
                                                       const float* __restrict__ dist, int cols, int64_t rows,
                                                       int min_points, float* __restrict__ out,
                                                       int32_t* __restrict__ argrow) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t v = gid / cout;
  const int c = (int)(gid - v * cout);
  if (v >= nv) return;
  int count = seg_start[v + 1] - seg_start[v];
  if (v == 0) count += seg_start[nv + 1] - seg_start[nv];  // folded rows (index -1)
  const unsigned long long p = packed[gid];
  if (p != 0ull) const_cast<unsigned long long*>(packed)[gid] = 0ull;   // the accumulators are zero between calls
  float val = 0.0f;          // torch_scatter: empty segment -> 0
  int64_t arg = rows;        // torch_scatter: empty segment -> src.size(0)
  if (p != 0ull) {
    val = tln_ord2f((uint32_t)(p >> 32));
    arg = (int64_t)(0xFFFFFFFFu - (uint32_t)(p & 0xFFFFFFFFull));
  }
  // row whose MLP output IS the pooled value (for the backward pass): -1 when the segment is empty or masked
  if (argrow) argrow[gid] = (p != 0ull && count >= min_points) ? (int32_t)arg : -1;
  if (arg > (int64_t)nv) arg = 0;  // lm:514 compares row ids against the number of vertices
  if (arg >= rows) arg = 0;        // (the reference would raise an index error here: rows <= V)
  float bary = dist[arg * cols + (cols - 1)];
  if (count < min_points) {
    val = 0.0f;
    bary = 0.0f;
  }
  out[v * (2 * cout) + c] = val;
  out[v * (2 * cout) + cout + c] = bary;
}

// ---------------------------------------------------------------------------------------
// The same pool on the VERTEX BINS of the distribute of this frame (lattice.hip, k_bins_*): the rows of a vertex are
// contiguous in the bin arrays, so a wave streams 64 x (16 + 4 + 4 + 4) contiguous bytes instead of gathering 20-byte
// rows through a sorted index list, and subtracts the vertex's local mean on the way (the [4N,5] `distributed`
// tensor is never read).  A vertex whose rows all lie inside the wave's 64-row chunk — three quarters of them — is
// FINISHED here: its output row (max, barycentric weight of the arg-max row with the lm:514 clamp, < min_points
// mask) is written directly; only vertices whose segment crosses a chunk boundary (and vertex 0, which also takes the
// rows without a vertex, lm:480) go through the packed 64-bit atomicMax and k_pool_bins_finalize.
// The rows of a segment arrive in arbitrary order: ties go to the smallest row id explicitly.
// ---------------------------------------------------------------------------------------
// The pool of up to TLN_POOL_MAXJOBS frames in one launch (blockIdx.y = frame): the lock-stepped sequences of a stream
// share the PointNet weights, only the bins and the outputs differ
#define TLN_POOL_MAXJOBS 8
struct PoolJob {
  TlnBins bn;
  int64_t rows;
  unsigned long long* packed;
  float* out;
  int32_t* argrow;
  int nv;
};
struct PoolJobs {
  PoolJob j[TLN_POOL_MAXJOBS];
};

template <int CIN, int H1, int H2, int COUT>
__global__ void __launch_bounds__(256) k_pool_bins(const PoolJobs jobs, int min_points,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                   const float* __restrict__ w3, const float* __restrict__ b3) {
  const PoolJob& J = jobs.j[blockIdx.y];
  const TlnBins& bn = J.bn;
  const int64_t rows = J.rows;
  unsigned long long* __restrict__ packed = J.packed;
  float* __restrict__ out = J.out;
  int32_t* __restrict__ argrow = J.argrow;
  constexpr int HL = (H1 == 0) ? CIN : (H2 ? H2 : H1);
  constexpr int TS = ((HL + 3) / 4) * 4 + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tile = smem + wid * (64 * TS + 64);
  float* tw = tile + 64 * TS;   // [64] barycentric weight of the chunk's rows

  const int64_t chunk = (int64_t)blockIdx.x * 4 + wid;
  const int64_t j0 = chunk * 64;
  if (j0 >= rows) return;  // no block barrier below: the tile is private to the wave
  const int cnt = (int)((rows - j0) < 64 ? (rows - j0) : 64);
  const int nv = bn.ctr[0];
  const float w0 = bn.weights[0];   // lm:514: an arg-max row id > V reads the barycentric weight of row 0

  // ---- phase 1: lane = row.  Vertex, row id and flags of the lane's row stay in registers: phase 2 fetches those of
  // row j with v_readlane (j is wave-uniform) — scalars, so the walk below branches on the scalar unit and keeps its
  // per-vertex state there, instead of comparing broadcast LDS values lane by lane under exec masks
  int my_v = 0, my_r = 0, my_f = 0;
  if (lane < cnt) {
    const int64_t at = j0 + lane;
    float4 q = bn.rec[at].a;
    const uint4 meta = bn.rec[at].m;
    const int v = (int)meta.z;
    int flags = 0, vv = 0;      // rows without a vertex fold into vertex 0 with their raw position (lm:480)
    if (v >= 0) {
      vv = v;
      if (bn.subtract) {
        q.x -= bn.mean[3 * v];
        q.y -= bn.mean[3 * v + 1];
        q.z -= bn.mean[3 * v + 2];
      }
      if (v != 0) {
        const int st = bn.vstart[v], c = bn.vcnt[v];
        if ((int64_t)st == at) flags |= 1;            // first row of its vertex
        if ((int64_t)st + c == at + 1) flags |= 2;    // last row of its vertex
        if (c < min_points) flags |= 4;
      }
    }
    const float xin[4] = {q.x, q.y, q.z, q.w};
    float x[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) x[c] = xin[c];
    float hl[HL];
    if constexpr (H1 == 0) {
#pragma unroll
      for (int c = 0; c < HL; ++c) hl[c] = x[c];
    } else if constexpr (H2 == 0) {
      dense<CIN, H1, true>(w1, b1, x, hl);
    } else {
      float h1[H1];
      dense<CIN, H1, true>(w1, b1, x, h1);
      dense<H1, H2, true>(w2, b2, h1, hl);
    }
#pragma unroll
    for (int c = 0; c < HL; ++c) tile[lane * TS + c] = hl[c];
    my_v = vv;
    my_r = (int)meta.y;
    my_f = flags;
    tw[lane] = __uint_as_float(meta.x);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2: lane = output channel of the last layer
  const int c = lane;
  const bool active = c < COUT;
  float wl[HL];
  float bias = 0.0f;
  if constexpr (H1 != 0) {
    const float* wlast = H2 ? w3 : w2;
    const float* blast = H2 ? b3 : b2;
#pragma unroll
    for (int i = 0; i < HL; ++i) wl[i] = active ? wlast[c * HL + i] : 0.0f;
    bias = active ? blast[c] : 0.0f;
  }
  auto value_of = [&](int j) {
    const float* h = tile + j * TS;
    if constexpr (H1 == 0) {
      return h[active ? c : 0];
    } else {
      float acc = bias;   // ONE fma chain in ascending i (oracle/csrc/pool_mlp.c)
#pragma unroll
      for (int i = 0; i < HL; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(h + i);   // one address for the whole wave: broadcast
        acc = fmaf(wl[i], q.x, acc);
        acc = fmaf(wl[i + 1], q.y, acc);
        acc = fmaf(wl[i + 2], q.z, acc);
        acc = fmaf(wl[i + 3], q.w, acc);
      }
      return acc;
    }
  };
  // a vertex's rows [js, je] of this chunk are done: its result goes straight to the output when they are ALL its rows
  auto flush = [&](int v, int js, int je, float best, int bj, int arg) {
    const int fs = __builtin_amdgcn_readlane(my_f, js), fe = __builtin_amdgcn_readlane(my_f, je);
    const bool whole = (fs & 1) && (fe & 2);   // wave-uniform
    if (!active) return;
    if (whole) {
      const bool masked = (fs & 4) != 0;
      const float bary = arg > nv ? w0 : tw[bj];
      out[(int64_t)v * (2 * COUT) + c] = masked ? 0.0f : best;
      out[(int64_t)v * (2 * COUT) + COUT + c] = masked ? 0.0f : bary;
      if (argrow) argrow[(int64_t)v * COUT + c] = masked ? -1 : arg;
    } else {
      const unsigned long long p = ((unsigned long long)tln_f2ord(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)arg);
      atomicMax(&packed[(int64_t)v * COUT + c], p);
    }
  };
  int cur = __builtin_amdgcn_readlane(my_v, 0), js = 0;   // scalars
  int bj = 0, brow = __builtin_amdgcn_readlane(my_r, 0);  // per lane (= channel): row of the running maximum
  float best = value_of(0);
  for (int j = 1; j < cnt; ++j) {
    const int v = __builtin_amdgcn_readlane(my_v, j);
    const int rj = __builtin_amdgcn_readlane(my_r, j);
    const float val = value_of(j);
    if (v != cur) {  // scalar branch
      flush(cur, js, j - 1, best, bj, brow);
      cur = v;
      js = j;
      best = val;
      bj = j;
      brow = rj;
    } else {
      const bool better = val > best || (val == best && rj < brow);   // ties: the smallest row id
      best = better ? val : best;
      bj = better ? j : bj;
      brow = better ? rj : brow;
    }
  }
  flush(cur, js, cnt - 1, best, bj, brow);
}

// ---------------------------------------------------------------------------------------
// k_pool_bins with TWO rows per step (round 4; the 4/3-16-32-64 MLP only).  k_pool_bins's last layer has lane = channel and
// fetches the 32 activations of ONE row per step by eight broadcast ds_read_b128 — four LDS cycles each whatever the number
// of distinct addresses per lane group, 32 LDS cycles per row and wave, the CU's LDS the busiest unit of the launch.  A
// ds_read_b128 serves its four groups of 16 lanes one address each, and every group lies inside one half of the wave: with
// the lower half on row t and the upper half on row 32 + t of the chunk the same eight reads deliver TWO rows.  Each lane
// then owns two channels (c and c + 32: two weight rows in registers, two fma chains per step — the same 32 fmas per row
// and wave as before), and each half walks its 32-row block by itself: vertex, row id and flags of a row come from small LDS
// arrays (one address per half), the running maxima and the flush are per lane.  A vertex whose rows cross the block
// boundary is not "whole" for either half — decided from the first-row / last-row flags as before — and goes through the
// packed atomicMax and k_pool_bins_finalize (which tests block boundaries of 32 rows for this kernel).  Same chains, same
// tie rule: bit for bit k_pool_bins.
template <int CIN>
__global__ void __launch_bounds__(256) k_pool_bins2(const PoolJobs jobs, int min_points, const float* __restrict__ w1,
                                                    const float* __restrict__ b1, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, const float* __restrict__ w3,
                                                    const float* __restrict__ b3) {
  constexpr int H1 = 16, H2 = 32, COUT = 64, HL = 32, TS = 36;
  const PoolJob& J = jobs.j[blockIdx.y];
  const TlnBins& bn = J.bn;
  const int64_t rows = J.rows;
  unsigned long long* __restrict__ packed = J.packed;
  float* __restrict__ out = J.out;
  int32_t* __restrict__ argrow = J.argrow;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tile = smem + wid * (64 * TS + 4 * 64);
  float* tw = tile + 64 * TS;                       // [64] barycentric weight of the chunk's rows
  int* tv = reinterpret_cast<int*>(tw + 64);        // [64] vertex
  int* tr = tv + 64;                                // [64] row id
  int* tf = tr + 64;                                // [64] flags
  const int64_t chunk = (int64_t)blockIdx.x * 4 + wid;
  const int64_t j0 = chunk * 64;
  if (j0 >= rows) return;  // no block barrier below: the tile is private to the wave
  const int cnt = (int)((rows - j0) < 64 ? (rows - j0) : 64);
  const int nv = bn.ctr[0];
  const float w0 = bn.weights[0];   // lm:514

  // ---- phase 1: lane = row (as k_pool_bins)
  if (lane < cnt) {
    const int64_t at = j0 + lane;
    float4 q = bn.rec[at].a;
    const uint4 meta = bn.rec[at].m;
    const int v = (int)meta.z;
    int flags = 0, vv = 0;
    if (v >= 0) {
      vv = v;
      if (bn.subtract) {
        q.x -= bn.mean[3 * v];
        q.y -= bn.mean[3 * v + 1];
        q.z -= bn.mean[3 * v + 2];
      }
      if (v != 0) {
        const int st = bn.vstart[v], c = bn.vcnt[v];
        if ((int64_t)st == at) flags |= 1;
        if ((int64_t)st + c == at + 1) flags |= 2;
        if (c < min_points) flags |= 4;
      }
    }
    const float xin[4] = {q.x, q.y, q.z, q.w};
    float x[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) x[c] = xin[c];
    float h1[H1], hl[HL];
    dense<CIN, H1, true>(w1, b1, x, h1);
    dense<H1, H2, true>(w2, b2, h1, hl);
#pragma unroll
    for (int c = 0; c < HL; ++c) tile[lane * TS + c] = hl[c];
    tw[lane] = __uint_as_float(meta.x);
    tv[lane] = vv;
    tr[lane] = (int)meta.y;
    tf[lane] = flags;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2: lane = (block of 32 rows, channels cl and cl + 32)
  const int half = lane >> 5, cl = lane & 31;
  float wa[HL], wb[HL];
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    wa[i] = w3[cl * HL + i];
    wb[i] = w3[(cl + 32) * HL + i];
  }
  const float bias_a = b3[cl], bias_b = b3[cl + 32];
  const int base = half * 32;
  const int mine = cnt - base < 0 ? 0 : (cnt - base > 32 ? 32 : cnt - base);   // rows of this half's block
  const int steps = cnt < 32 ? cnt : 32;                                       // (wave-uniform: the lower block's rows)
  // one vertex segment of this half: (cur, flags of its first row, flags of its latest row), the two running maxima
  int cur = -1, f_first = 0, f_last = 0;
  float best_a = 0.0f, best_b = 0.0f;
  int bj_a = 0, bj_b = 0, brow_a = 0, brow_b = 0;
  auto flush = [&]() {
    const bool whole = (f_first & 1) && (f_last & 2);
    if (whole) {
      const bool masked = (f_first & 4) != 0;
      const float bary_a = brow_a > nv ? w0 : tw[bj_a];
      const float bary_b = brow_b > nv ? w0 : tw[bj_b];
      float* o = out + (int64_t)cur * (2 * COUT);
      o[cl] = masked ? 0.0f : best_a;
      o[cl + 32] = masked ? 0.0f : best_b;
      o[COUT + cl] = masked ? 0.0f : bary_a;
      o[COUT + cl + 32] = masked ? 0.0f : bary_b;
      if (argrow) {
        argrow[(int64_t)cur * COUT + cl] = masked ? -1 : brow_a;
        argrow[(int64_t)cur * COUT + cl + 32] = masked ? -1 : brow_b;
      }
    } else {
      const unsigned long long pa = ((unsigned long long)tln_f2ord(best_a) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)brow_a);
      const unsigned long long pb = ((unsigned long long)tln_f2ord(best_b) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)brow_b);
      atomicMax(&packed[(int64_t)cur * COUT + cl], pa);
      atomicMax(&packed[(int64_t)cur * COUT + cl + 32], pb);
    }
  };
  for (int t = 0; t < steps; ++t) {
    const bool valid = t < mine;
    const int j = valid ? base + t : 0;        // (a half without this row reads row 0 and drops the result)
    const float* h = tile + j * TS;
    float va = bias_a, vb = bias_b;            // ONE fma chain per channel in ascending i
#pragma unroll
    for (int i = 0; i < HL; i += 4) {
      const float4 q = *reinterpret_cast<const float4*>(h + i);   // one address per half of the wave
      va = fmaf(wa[i], q.x, va);
      vb = fmaf(wb[i], q.x, vb);
      va = fmaf(wa[i + 1], q.y, va);
      vb = fmaf(wb[i + 1], q.y, vb);
      va = fmaf(wa[i + 2], q.z, va);
      vb = fmaf(wb[i + 2], q.z, vb);
      va = fmaf(wa[i + 3], q.w, va);
      vb = fmaf(wb[i + 3], q.w, vb);
    }
    const int v = tv[j], rj = tr[j], f = tf[j];
    if (valid) {
      if (v != cur) {
        if (cur >= 0) flush();
        cur = v;
        f_first = f;
        best_a = va;
        best_b = vb;
        bj_a = bj_b = j;
        brow_a = brow_b = rj;
      } else {
        const bool ga = va > best_a || (va == best_a && rj < brow_a);   // ties: the smallest row id
        const bool gb = vb > best_b || (vb == best_b && rj < brow_b);
        best_a = ga ? va : best_a;
        bj_a = ga ? j : bj_a;
        brow_a = ga ? rj : brow_a;
        best_b = gb ? vb : best_b;
        bj_b = gb ? j : bj_b;
        brow_b = gb ? rj : brow_b;
      }
      f_last = f;
    }
  }
  if (cur >= 0) flush();
}

// ---------------------------------------------------------------------------------------
// The 4-16-32-64 pool with its two wide layers on the matrix cores AND the per-vertex max taken where the accumulator
// leaves the values (round 3).  v_mfma_f32_32x32x2_f32 accumulates like a chain of fp32 fmas in ascending k
// (tools/micro/mfma_chain.hip), so C = bias followed by K/2 MFMAs reproduces the pinned summation order
// bit for bit.  legacy.hip's first matrix-core version wrote every 32 x 64 result tile to LDS and walked it row by row
// with lane = channel — ~2000 vector instructions per 64 rows, as many cycles as the MFMAs — and was no faster than the
// all-VALU kernel.  Here:
//   layer 1 (4 -> 16)   lane = row, VALU (64 fmas)
//   layer 2 (16 -> 32)  D[o][row], A = W2, B = h1 of a 32-row tile (one v_permlane32_swap per k pair serves both tiles)
//   layer 3 (32 -> 64)  D[row][o], A = h2 from layer 2's accumulator, B = W3: a lane ends up with 16 rows
//                       {8g + 4h + m} of ONE channel (h = lane / 32)
//   relayout            8 v_permlane32_swap per tile and column half: the lower lanes now hold rows 0-15 of the tile, the
//                       upper lanes rows 16-31, in order: a 64-row chunk is four BLOCKS of 16 consecutive rows
//   max / arg-max       every lane walks its block's 16 values in registers (a run of equal vertices starts where the
//                       wave-uniform start mask says; ties go to the smallest row id): runs inside a block are flushed on
//                       the spot, a run that crosses a block boundary hands (value, row) to the other half of the wave
//                       (three hand-offs per chunk) — ~13 vector instructions per row for TWO channels.
// Flush = what k_pool_bins does: a vertex whose rows all lie inside the chunk is written straight to the output, the
// others (and vertex 0) go through the packed 64-bit atomicMax and k_pool_bins_finalize.
// ---------------------------------------------------------------------------------------
typedef float f16p __attribute__((ext_vector_type(16)));
typedef unsigned u2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pswap32(float x, float y, float& lo, float& hi) {
  // lo = {x[0:31], y[0:31]},  hi = {x[32:63], y[32:63]}
  const u2p r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  lo = __uint_as_float(r.x);
  hi = __uint_as_float(r.y);
}

template <int CIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_pool_bins_mx(
    const PoolJobs jobs, int min_points, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3) {
  const PoolJob& J = jobs.j[blockIdx.y];
  const TlnBins& bn = J.bn;
  const int64_t rows = J.rows;
  unsigned long long* __restrict__ packed = J.packed;
  float* __restrict__ out = J.out;
  int32_t* __restrict__ argrow = J.argrow;
  constexpr int H1 = 16, H2 = 32, COUT = 64;
  __shared__ int tv_s[4][64], trow_s[4][64];     // vertex (-2: no row) and row id of the chunk's rows, per wave
  __shared__ float tw_s[4][64];                  // their barycentric weights (the arg-max row's is part of the output)
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l32 = lane & 31;
  int* tv = tv_s[wid];
  int* trow = trow_s[wid];
  float* tw = tw_s[wid];

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
    r.v = -2;                       // no row here (past the frame's rows)
    r.row = 0;
    r.w = 0.0f;
    r.q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < chunks && at < rows) {
      r.q = bn.rec[at].a;
      const uint4 meta = bn.rec[at].m;
      r.v = (int)meta.z;            // -1: a row without a vertex
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

  const float NEG = -__builtin_inff();
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
    int my_v = -2;                  // rows without a vertex fold into vertex 0 with their raw position (lm:480)
    bool r_whole = false, r_masked = false;
    if (rin.v != -2) {
      my_v = 0;
      float4 q = rin.q;
      if (rin.v >= 0) {
        my_v = rin.v;
        q.x -= vin.mx;
        q.y -= vin.my;
        q.z -= vin.mz;
        r_whole = rin.v != 0 && (int64_t)vin.st >= j0 && (int64_t)vin.st + vin.c <= j0 + cnt;
        r_masked = vin.c < min_points;
      }
      const float xin[4] = {q.x, q.y, q.z, q.w};
      float x[CIN];
#pragma unroll
      for (int k = 0; k < CIN; ++k) x[k] = xin[k];
      dense<CIN, H1, true>(w1, b1, x, h1);
    }
    tv[lane] = my_v;
    trow[lane] = rin.row;
    tw[lane] = rin.w;
    // wave-uniform masks over the 64 rows: first row of every run of equal vertices (rows past the end: one dead run)
    const int v_prev = __shfl_up(my_v, 1, 64);
    const unsigned long long startmask = __ballot(lane == 0 || my_v != v_prev);
    const unsigned long long wholemask = __ballot(r_whole);
    const unsigned long long maskedmask = __ballot(r_masked);
    // ---- layer 2 on both 32-row tiles
    f16p acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc0[i] = bias2[i];
      acc1[i] = bias2[i];
    }
#pragma unroll
    for (int j = 0; j < H1 / 2; ++j) {
      float t0, t1;
      pswap32(h1[2 * j], h1[2 * j + 1], t0, t1);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], t0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], t1, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): tv / trow of this chunk are in LDS
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // a finished run (its last row at chunk position `pos`): straight to the output if the vertex lies inside the chunk
    auto flush_run = [&](int pos, const float (&val)[2], const int (&row)[2], const int (&at)[2]) {
      const int v = tv[pos];
      if (v == -2) return;          // the dead run behind the frame's last row
      const bool whole = (wholemask >> pos) & 1ull;
      const bool masked = (maskedmask >> pos) & 1ull;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c = l32 + 32 * nt;
        if (whole) {
          const float bary = row[nt] > nv ? w0 : tw[at[nt]];   // lm:514 (the arg-max row's weight: from LDS, by position)
          out[(int64_t)v * (2 * COUT) + c] = masked ? 0.0f : val[nt];
          out[(int64_t)v * (2 * COUT) + COUT + c] = masked ? 0.0f : bary;
          if (argrow) argrow[(int64_t)v * COUT + c] = masked ? -1 : row[nt];
        } else {
          const unsigned long long p = ((unsigned long long)tln_f2ord(val[nt]) << 32) |
                                       (unsigned long long)(0xFFFFFFFFu - (uint32_t)row[nt]);
          atomicMax(&packed[(int64_t)v * COUT + c], p);
        }
      }
    };
    auto better = [](float v, int r, float bv, int br) { return v > bv || (v == bv && r < br); };

    // the run open at the end of the previous block, in the lanes of THIS half once handed over
    float cval[2] = {NEG, NEG};
    int crow[2] = {0x7fffffff, 0x7fffffff}, cat[2] = {0, 0};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // ---- layer 3 for the tile: 2 x 16 MFMAs, then the rows of a channel in order (lower lanes 0-15, upper 16-31)
      f16p h2 = t ? acc1 : acc0;
#pragma unroll
      for (int i = 0; i < 16; ++i) h2[i] = fmaxf(h2[i], 0.0f);
      float aop[H2 / 2];
#pragma unroll
      for (int jj = 0; jj < H2 / 2; ++jj) {
        if ((jj & 3) < 2) {   // the k pairs whose channels sit in the lower half's registers
          const int i0 = 4 * (jj / 4) + 2 * (jj % 2);
          pswap32(h2[i0], h2[i0 + 1], aop[jj], aop[jj + 2]);
        }
      }
      float n[2][16];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f16p d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = bias3[nt];
#pragma unroll
        for (int j = 0; j < H2 / 2; ++j) d = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[j], bw3[nt][j], d, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          pswap32(d[m], d[8 + m], n[nt][m], n[nt][4 + m]);
          pswap32(d[4 + m], d[12 + m], n[nt][8 + m], n[nt][12 + m]);
        }
      }
      // ---- the block's walk
      const int base = 32 * t + 16 * half;
      const unsigned tile_bits = (unsigned)(startmask >> (32 * t));
      const unsigned sb = half ? (tile_bits >> 16) : (tile_bits & 0xFFFFu);    // start bits of this lane's 16 rows
      bool seen = false;
      float best[2] = {NEG, NEG}, headv[2] = {NEG, NEG};
      int brow[2] = {0x7fffffff, 0x7fffffff}, headr[2] = {0x7fffffff, 0x7fffffff};
      int bat[2] = {0, 0}, heada[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bool st = (sb >> j) & 1u;
        const int rj = trow[base + j];
        if (st) {
          if (seen) {
            flush_run(base + j - 1, best, brow, bat);      // a run that began and ended inside this block
          } else {
            headv[0] = best[0];
            headv[1] = best[1];
            headr[0] = brow[0];
            headr[1] = brow[1];
            heada[0] = bat[0];
            heada[1] = bat[1];
            seen = true;
          }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float v = n[nt][j];
          const bool take = st || better(v, rj, best[nt], brow[nt]);
          best[nt] = take ? v : best[nt];
          brow[nt] = take ? rj : brow[nt];
          bat[nt] = take ? base + j : bat[nt];
        }
      }
      // ---- stitch the blocks: lower half (block 2t) first, its open run goes to the upper half (block 2t + 1), and
      // that one's to the lower half of the next tile.  All conditions but `seen` are wave-uniform.
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int b = 2 * t + hb;                                   // block being finished
        const bool cont_in = b > 0 && !((startmask >> (16 * b)) & 1ull);         // a run continues into it
        const bool ends = b == 3 || ((startmask >> (16 * (b + 1))) & 1ull);      // its last run ends with it
        if (b > 0) {   // the carry of block b - 1 sits in the other half's lanes
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float xv = __shfl_xor(cval[nt], 32, 64);
            const int xr = __shfl_xor(crow[nt], 32, 64);
            const int xa = __shfl_xor(cat[nt], 32, 64);
            if (half == hb) {
              cval[nt] = xv;
              crow[nt] = xr;
              cat[nt] = xa;
            }
          }
        }
        if (half == hb) {
          float outv[2];
          int outr[2], outa[2];
          if (seen) {
            if (cont_in) {   // the older run ends inside this block: carry + head
              float hv[2];
              int hr[2], ha[2];
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                const bool tk = better(headv[nt], headr[nt], cval[nt], crow[nt]);
                hv[nt] = tk ? headv[nt] : cval[nt];
                hr[nt] = tk ? headr[nt] : crow[nt];
                ha[nt] = tk ? heada[nt] : cat[nt];
              }
              flush_run(base + (__ffs((int)sb) - 1) - 1, hv, hr, ha);
            }
            outv[0] = best[0];
            outv[1] = best[1];
            outr[0] = brow[0];
            outr[1] = brow[1];
            outa[0] = bat[0];
            outa[1] = bat[1];
          } else {           // no run starts in this block: it is all continuation
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const bool tk = !cont_in || better(best[nt], brow[nt], cval[nt], crow[nt]);
              outv[nt] = tk ? best[nt] : cval[nt];
              outr[nt] = tk ? brow[nt] : crow[nt];
              outa[nt] = tk ? bat[nt] : cat[nt];
            }
          }
          if (ends) flush_run(base + 15, outv, outr, outa);
          cval[0] = outv[0];
          cval[1] = outv[1];
          crow[0] = outr[0];
          crow[1] = outr[1];
          cat[0] = outa[0];
          cat[1] = outa[1];
        }
      }
    }
    load_vtx(rnext, vnext);   // the next chunk's rows have arrived by now: what their vertices point to
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // tv / trow are rewritten by the next chunk
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    rin = rnext;
    vin = vnext;
  }
}

// what k_pool_bins left open: vertices without rows (zeros, torch_scatter's empty segment), vertices whose segment
// crosses a 64-row chunk boundary and vertex 0 (packed accumulators -> value, barycentric weight, mask)
__global__ void __launch_bounds__(256) k_pool_bins_finalize(const PoolJobs jobs, int cout, int min_points, int block_shift) {
  const PoolJob& J = jobs.j[blockIdx.y];
  const TlnBins& bn = J.bn;
  const int nv = J.nv;
  const int64_t rows = J.rows;
  unsigned long long* packed = J.packed;
  float* __restrict__ out = J.out;
  int32_t* __restrict__ argrow = J.argrow;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t v = gid / cout;
  const int c = (int)(gid - v * cout);
  if (v >= nv) return;
  const bool visited = bn.vstamp == nullptr || bn.vstamp[v] == bn.stamp;
  const int st = visited ? bn.vstart[v] : 0;
  int count = visited ? bn.vcnt[v] : 0;
  // (block_shift: 6 = k_pool_bins's 64-row chunks, 5 = the 32-row blocks of k_pool_bins2)
  const bool crossing = v == 0 || (count > 0 && (st >> block_shift) != ((st + count - 1) >> block_shift));
  if (v == 0) count += bn.ctr[2];                  // the rows without a vertex (counted by K1) fold into vertex 0
  if (count > 0 && !crossing) return;              // written by k_pool_bins
  float val = 0.0f, bary = 0.0f;
  int32_t argout = -1;
  if (count > 0) {
    const unsigned long long p = packed[gid];
    packed[gid] = 0ull;                            // zero between calls
    int64_t arg = rows;
    if (p != 0ull) {
      val = tln_ord2f((uint32_t)(p >> 32));
      arg = (int64_t)(0xFFFFFFFFu - (uint32_t)(p & 0xFFFFFFFFull));
    }
    if (p != 0ull && count >= min_points) argout = (int32_t)arg;
    if (arg > (int64_t)nv) arg = 0;   // lm:514
    if (arg >= rows) arg = 0;
    bary = bn.weights[arg];
    if (count < min_points) {
      val = 0.0f;
      bary = 0.0f;
    }
  }
  if (argrow) argrow[gid] = argout;
  out[v * (2 * cout) + c] = val;
  out[v * (2 * cout) + cout + c] = bary;
}

template <int CIN, int H1, int H2, int COUT>
static int launch_pool_bins(const PoolJobs& jobs, int n, const float* const* w, const float* const* b, int min_points,
                            hipStream_t s) {
  constexpr int HL = (H1 == 0) ? CIN : (H2 ? H2 : H1);
  constexpr int TS = ((HL + 3) / 4) * 4 + 4;
  const size_t lds = (size_t)(4 * (64 * TS + 64)) * sizeof(float);   // 37.9 KB for the 32-wide last hidden layer: four workgroups per CU
  MlpParams mp{};
  for (int i = 0; i < 3; ++i) {
    mp.w[i] = (H1 && w) ? w[i < (H2 ? 3 : 2) ? i : 0] : nullptr;
    mp.b[i] = (H1 && b) ? b[i < (H2 ? 3 : 2) ? i : 0] : nullptr;
  }
  int64_t rows = 0;
  for (int i = 0; i < n; ++i)
    if (jobs.j[i].rows > rows) rows = jobs.j[i].rows;   // the grid is sized for the longest frame
  const int64_t chunks = tln_cdiv(rows, 64);
  auto kern = k_pool_bins<CIN, H1, H2, COUT>;
  static thread_local TlnLdsAttr attr;   // (one per template instantiation)
  TLN_HIP(tln_set_max_lds(attr, reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)tln_cdiv(chunks, 4), (unsigned)n), dim3(256), lds, s, jobs, min_points, mp.w[0],
                     mp.b[0], mp.w[1], mp.b[1], mp.w[2], mp.b[2]);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

template <int CIN, int H1, int H2, int COUT>
static int launch_pool(tln_lattice* l, const float* d_dist, int64_t rows, int cols, const float* const* w,
                       const float* const* b, int nv, unsigned long long* packed, hipStream_t s) {
  constexpr int HL = (H1 == 0) ? CIN : (H2 ? H2 : H1);
  constexpr int TS = ((HL + 3) / 4) * 4 + 4;
  const size_t lds = (size_t)(4 * (64 * TS + 128)) * sizeof(float);
  MlpParams mp{};
  for (int i = 0; i < 3; ++i) {
    mp.w[i] = (H1 && w) ? w[i < (H2 ? 3 : 2) ? i : 0] : nullptr;
    mp.b[i] = (H1 && b) ? b[i < (H2 ? 3 : 2) ? i : 0] : nullptr;
  }
  const int64_t chunks = tln_cdiv(rows, 64);
  auto kern = k_pool_chunks<CIN, H1, H2, COUT>;
  static thread_local TlnLdsAttr attr;   // (one per template instantiation)
  TLN_HIP(tln_set_max_lds(attr, reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)tln_cdiv(chunks, 4)), dim3(256), lds, s, d_dist, cols, tln_lat_order(l),
                     tln_lat_sorted_vertex(l), rows, nv, mp.w[0], mp.b[0], mp.w[1], mp.b[1], mp.w[2], mp.b[2], packed);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

#define TLN_POOL_DEFAULT_MODE 0
// which kernel pools the 4-16-32-64 MLP from the bins: the all-VALU k_pool_bins (default) or k_pool_bins_mfma
// (TLN_POOL_MFMA=1 / tln_options.pool_mode = 1).  Bitwise equal results; the MFMA variant measured no faster.
// (tln_options.pool_mode of the lattice handle, tln_lattice_set_options; -1 = env TLN_POOL_MFMA, read once, else 0)
static int pool_mode(const tln_lattice* l) {   // 0: all-VALU k_pool_bins, 1: legacy.hip's matrix-core variant, 2: k_pool_bins_mx,
                                               // 3: k_pool_bins2 (two rows per step)
  static const int env_mode = [] {
    const char* e = getenv("TLN_POOL_MFMA");
    const int m = e ? atoi(e) : TLN_POOL_DEFAULT_MODE;
    return (m < 0 || m > 3) ? TLN_POOL_DEFAULT_MODE : m;
  }();
  const int m = tln_lat_options(l).pool_mode;
  return (m < 0 || m > 3) ? env_mode : m;
}

// the bins route for 1..TLN_POOL_MAXJOBS frames that share the MLP: *taken = false when the shape has no bins kernel
static int pool_bins_jobs(PoolJobs& jobs, int n, int nr_layers, const float* const* d_w, const float* const* d_b,
                          const int* dims, int min_points, bool* taken, int mode, hipStream_t s) {
  const int cin = dims[0], cout = dims[nr_layers];
  int rc = TLN_OK;
  for (int i = n; i < TLN_POOL_MAXJOBS; ++i) jobs.j[i] = jobs.j[0];   // (never indexed: the grids have n rows)
#define POOLB_CASE(CI, A, B, CO) rc = launch_pool_bins<CI, A, B, CO>(jobs, n, d_w, d_b, min_points, s)
  *taken = true;
  const bool shape_wide = nr_layers == 3 && dims[1] == 16 && dims[2] == 32 && dims[3] == 64 && (cin == 4 || cin == 3);
  const bool wide = n == 1 && shape_wide && mode == 1;
  const PoolJob& j0 = jobs.j[0];
  int block_shift = 6;
  if (shape_wide && mode == 3) {
    int64_t rmax = 0;
    for (int i = 0; i < n; ++i)
      if (jobs.j[i].rows > rmax) rmax = jobs.j[i].rows;
    const size_t lds = (size_t)(4 * (64 * 36 + 4 * 64)) * sizeof(float);   // 40 KB: four workgroups per CU
    const dim3 grid((unsigned)tln_cdiv(tln_cdiv(rmax, 64), 4), (unsigned)n);
    if (cin == 4) {
      static thread_local TlnLdsAttr attr4;
      TLN_HIP(tln_set_max_lds(attr4, reinterpret_cast<const void*>(k_pool_bins2<4>), (int)lds));
      hipLaunchKernelGGL(k_pool_bins2<4>, grid, dim3(256), lds, s, jobs, min_points, d_w[0], d_b[0], d_w[1], d_b[1], d_w[2], d_b[2]);
    } else {
      static thread_local TlnLdsAttr attr3;
      TLN_HIP(tln_set_max_lds(attr3, reinterpret_cast<const void*>(k_pool_bins2<3>), (int)lds));
      hipLaunchKernelGGL(k_pool_bins2<3>, grid, dim3(256), lds, s, jobs, min_points, d_w[0], d_b[0], d_w[1], d_b[1], d_w[2], d_b[2]);
    }
    TLN_LAUNCH_CHECK();
    block_shift = 5;
  } else if (shape_wide && mode == 2) {
    // matrix cores + the max in accumulator layout: a wave strides over chunks (weights stay in its registers)
    int64_t rmax = 0;
    for (int i = 0; i < n; ++i)
      if (jobs.j[i].rows > rmax) rmax = jobs.j[i].rows;
    int64_t blocks = tln_cdiv(tln_cdiv(rmax, 64), 4);
    const int64_t cap = tln_cdiv(512, n) < 64 ? 64 : tln_cdiv(512, n);   // two workgroups per CU over all jobs
    if (blocks > cap) blocks = cap;
    if (cin == 4)
      hipLaunchKernelGGL(k_pool_bins_mx<4>, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, s, jobs, min_points, d_w[0], d_b[0],
                         d_w[1], d_b[1], d_w[2], d_b[2]);
    else
      hipLaunchKernelGGL(k_pool_bins_mx<3>, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, s, jobs, min_points, d_w[0], d_b[0],
                         d_w[1], d_b[1], d_w[2], d_b[2]);
    TLN_LAUNCH_CHECK();
  } else if (wide && (cin == 4 || cin == 3))
    rc = tln_pool_bins_mfma_launch(cin, j0.bn, j0.rows, d_w, d_b, min_points, j0.packed, j0.out, j0.argrow, s);
  else if (nr_layers == 3 && cin == 4 && dims[1] == 16 && dims[2] == 32 && dims[3] == 64) POOLB_CASE(4, 16, 32, 64);
  else if (nr_layers == 2 && cin == 4 && dims[1] == 16 && dims[2] == 32) POOLB_CASE(4, 16, 0, 32);
  else if (nr_layers == 3 && cin == 3 && dims[1] == 16 && dims[2] == 32 && dims[3] == 64) POOLB_CASE(3, 16, 32, 64);
  else if (nr_layers == 0 && cin == 4) POOLB_CASE(4, 0, 0, 4);
  else if (nr_layers == 0 && cin == 3) POOLB_CASE(3, 0, 0, 3);
  else *taken = false;
#undef POOLB_CASE
  if (!*taken || rc) return rc;
  int nvmax = 0;
  for (int i = 0; i < n; ++i)
    if (jobs.j[i].nv > nvmax) nvmax = jobs.j[i].nv;
  const int64_t total = (int64_t)nvmax * cout;
  hipLaunchKernelGGL(k_pool_bins_finalize, dim3((unsigned)tln_cdiv(total, 256), (unsigned)n), dim3(256), 0, s, jobs, cout,
                     min_points, block_shift);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

extern "C" int tln_pointnet_pool_ex(tln_lattice_t* l, const float* d_distributed, int64_t rows, int dist_cols,
                                    int nr_layers, const float* const* d_w, const float* const* d_b, const int* dims,
                                    int min_points, float* d_out, int32_t* d_argrow, void* stream_) {
  TLN_REQUIRE(l && d_out && dims, "null argument");
  hipStream_t s = (hipStream_t)stream_;
  const int nv = (int)tln_lattice_nr_vertices(l);
  if (nv <= 0) return TLN_OK;
  const int cin = dims[0], cout = dims[nr_layers];
  TLN_REQUIRE(cin <= dist_cols, "MLP input %d wider than the distributed rows %d", cin, dist_cols);
  unsigned long long* packed = nullptr;
  int rc = tln_lat_pool_ws(l, (int64_t)nv * cout, &packed);
  if (rc) return rc;
  // the rows of this very frame's distribute: pool them from the vertex bins (no sorted row list, no [4N,5] reads)
  TlnBins bn;
  if (cin <= 4 && dist_cols == 5 && tln_lat_bins(l, d_distributed, rows, &bn)) {
    PoolJobs jobs;
    jobs.j[0] = PoolJob{bn, rows, packed, d_out, d_argrow, nv};
    bool taken = false;
    rc = pool_bins_jobs(jobs, 1, nr_layers, d_w, d_b, dims, min_points, &taken, pool_mode(l), s);
    if (taken) return rc;
  }
  TLN_REQUIRE(d_distributed, "the pool needs the distributed rows (or the bins of this frame's distribute)");
  TLN_REQUIRE(tln_lat_csr_rows(l) == rows, "pool needs the CSR of a tln_build_csr call over the same %lld rows",
              (long long)rows);
#define POOL_CASE(CI, A, B, CO) rc = launch_pool<CI, A, B, CO>(l, d_distributed, rows, dist_cols, d_w, d_b, nv, packed, s)
  if (nr_layers == 3 && cin == 4 && dims[1] == 16 && dims[2] == 32 && dims[3] == 64) POOL_CASE(4, 16, 32, 64);
  else if (nr_layers == 2 && cin == 4 && dims[1] == 16 && dims[2] == 32) POOL_CASE(4, 16, 0, 32);
  else if (nr_layers == 3 && cin == 3 && dims[1] == 16 && dims[2] == 32 && dims[3] == 64) POOL_CASE(3, 16, 32, 64);
  else if (nr_layers == 0 && cin == 4) POOL_CASE(4, 0, 0, 4);
  else if (nr_layers == 0 && cin == 3) POOL_CASE(3, 0, 0, 3);
  else {
    tln_set_error("unsupported PointNet shape: %d layers, cin %d (supported: 4-16-32-64, 4-16-32, 3-16-32-64, identity)",
                  nr_layers, cin);
    return TLN_E_INVALID;
  }
#undef POOL_CASE
  if (rc) return rc;
  const int64_t total = (int64_t)nv * cout;
  hipLaunchKernelGGL(k_pool_finalize, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, s, packed,
                     tln_lat_seg_start(l), nv, cout, d_distributed, dist_cols, rows, min_points, d_out, d_argrow);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}

extern "C" int tln_pointnet_pool(tln_lattice_t* l, const float* d_distributed, int64_t rows, int dist_cols,
                                 int nr_layers, const float* const* d_w, const float* const* d_b, const int* dims,
                                 int min_points, float* d_out, void* stream_) {
  return tln_pointnet_pool_ex(l, d_distributed, rows, dist_cols, nr_layers, d_w, d_b, dims, min_points, d_out, nullptr,
                              stream_);
}

// the pools of n lock-stepped sequences (same MLP, each lattice's own bins) as one launch per kernel; calls that cannot
// take the bins route (rows edited by the caller, unsupported shape) go through tln_pointnet_pool one by one
extern "C" int tln_pointnet_pool_multi(const tln_pool_call* c, int n, int dist_cols, int nr_layers, const float* const* d_w,
                                       const float* const* d_b, const int* dims, int min_points, void* stream_) {
  TLN_REQUIRE(c && n >= 1 && dims, "null argument");
  hipStream_t s = (hipStream_t)stream_;
  const int cin = dims[0], cout = dims[nr_layers];
  bool batch = n >= 2 && n <= TLN_POOL_MAXJOBS && cin <= 4 && dist_cols == 5;
  PoolJobs jobs;
  for (int i = 0; i < n && batch; ++i) {
    TLN_REQUIRE(c[i].l && c[i].d_out, "null argument");
    const int nv = (int)tln_lattice_nr_vertices(c[i].l);
    TlnBins bn;
    if (nv <= 0 || !tln_lat_bins(c[i].l, c[i].d_distributed, c[i].rows, &bn)) {
      batch = false;
      break;
    }
    unsigned long long* packed = nullptr;
    int rc = tln_lat_pool_ws(c[i].l, (int64_t)nv * cout, &packed);
    if (rc) return rc;
    jobs.j[i] = PoolJob{bn, c[i].rows, packed, c[i].d_out, nullptr, nv};
  }
  if (batch) {
    bool taken = false;
    int rc = pool_bins_jobs(jobs, n, nr_layers, d_w, d_b, dims, min_points, &taken, pool_mode(c[0].l), s);
    if (taken) return rc;
  }
  for (int i = 0; i < n; ++i) {
    int rc = tln_pointnet_pool(c[i].l, c[i].d_distributed, c[i].rows, dist_cols, nr_layers, d_w, d_b, dims, min_points,
                               c[i].d_out, stream_);
    if (rc) return rc;
  }
  return TLN_OK;
}

// ---------------------------------------------------------------------------------------
// K11 plain splat: out[v] = sum_rows w * [values, 1]   (wave per vertex, rows in sorted order,
// double accumulation with a fixed reduction tree => run-to-run reproducible)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_splat(const float* __restrict__ values, int val_dim,
                                               const float* __restrict__ weights, const int32_t* __restrict__ order,
                                               const int32_t* __restrict__ seg_start, int64_t nv,
                                               float* __restrict__ out) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (v >= nv) return;
  const int lane = threadIdx.x & 63;
  const int b = seg_start[v], e = seg_start[v + 1];
  for (int c = 0; c <= val_dim; ++c) {
    double acc = 0.0;
    for (int j = b + lane; j < e; j += 64) {
      const int64_t row = order[j];
      const float w = weights[row];
      const float x = (c < val_dim) ? values[(row >> 2) * val_dim + c] : 1.0f;
      acc += (double)(w * x);
    }
    acc = tln_wave_sum(acc);
    if (lane == 0) out[v * (val_dim + 1) + c] = (float)acc;
  }
}

extern "C" int tln_splat(tln_lattice_t* l, const float* d_values, int val_dim, const float* d_weights, int64_t rows,
                         float* d_out, void* stream_) {
  TLN_REQUIRE(l && d_weights && d_out && (val_dim == 0 || d_values), "null argument");
  TLN_REQUIRE(tln_lat_csr_rows(l) == rows, "splat needs the CSR of the last distribute over %lld rows", (long long)rows);
  const int64_t nv = tln_lattice_nr_vertices(l);
  if (nv <= 0) return TLN_OK;
  hipLaunchKernelGGL(k_splat, dim3((unsigned)tln_cdiv(nv * 64, 256)), dim3(256), 0, (hipStream_t)stream_, d_values,
                     val_dim, d_weights, tln_lat_order(l), tln_lat_seg_start(l), nv, d_out);
  TLN_LAUNCH_CHECK();
  return TLN_OK;
}
