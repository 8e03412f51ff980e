// Frame program interpreter (include/tln.h "Frame program"): the per-frame forward of LNN_SEQ
// (reference seq_lattice/models.py:284-476) driven from native code.  Host-only logic: every op resolves its
// slots to device pointers and calls the same C-ABI entry points the operator-level route calls
// (tln_gather_gemm_ex, tln_pointnet_pool, tln_gru_cell, tln_aflow, tln_slice...), so the two routes launch
// identical kernels with identical arguments.
//
// Memory: temporaries live in ONE arena owned by the program.  Before a frame is launched the op list is walked
// once without launching anything to find the arena high-water mark for this frame's vertex counts (the walk is
// deterministic, the real pass repeats it), the arena grows if needed, then the ops are launched.  A temporary is
// released right after the last op that reads it; everything is stream-ordered, so reuse needs no events.
// Hidden states sit in two buffers per state (previous / new) that swap when the frame that wrote one ends.
#include <map>
#include <vector>

#include "common.h"

// strided row copy (concat halves, the PointNet clone); zero_row0 folds the lm:569-570 row-0 reset into the clone
__global__ void __launch_bounds__(256) k_copy_rows(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst,
                                                   int64_t ld_dst, int64_t rows, int cols, int zero_row0) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = cols >> 2;  // cols % 4 == 0 and 16-byte aligned rows (checked by the caller)
  const int64_t r = gid / c4;
  if (r >= rows) return;
  const int c = (int)(gid - r * c4) * 4;
  float4 v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
  if (zero_row0 && r == 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(dst + r * ld_dst + c) = v;
}

// the same for up to eight copies of one width (lock-stepped sequences): blockIdx.y = copy
struct CopyJobs {
  struct {
    const float* src;
    float* dst;
    int64_t ld_src, ld_dst, rows;
    int zero_row0;
  } j[8];
};
__global__ void __launch_bounds__(256) k_copy_rows_multi(const CopyJobs jobs, int cols) {
  const auto& J = jobs.j[blockIdx.y];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = cols >> 2;
  const int64_t r = gid / c4;
  if (r >= J.rows) return;
  const int c = (int)(gid - r * c4) * 4;
  float4 v = *reinterpret_cast<const float4*>(J.src + r * J.ld_src + c);
  if (J.zero_row0 && r == 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(J.dst + r * J.ld_dst + c) = v;
}

// group mode: bit k of tln_options.group_off_mask set = ops of kind k (TLN_OP_*) are NOT batched over the group but
// launched per program, as before round 3 (test / measurement switch; env TLN_GROUP_BATCH_OFF, read once, is OR-ed in).
// Bit 0: the K1 / coarse-level batches of tln_program_begin_frame_group and the table batch of tln_program_run_group.
// A group follows its first program's options.
static inline bool group_batches(const tln_options& o, int kind) {
  static const int env_off = getenv("TLN_GROUP_BATCH_OFF") ? atoi(getenv("TLN_GROUP_BATCH_OFF")) : 0;
  return (((o.group_off_mask | env_off) >> kind) & 1) == 0;
}

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t b) { return (b + kAlign - 1) / kAlign * kAlign; }

// first-fit allocator over [0, capacity) with coalescing; offsets only, so it also serves the dry pass
struct Arena {
  std::map<size_t, size_t> free_;  // offset -> size
  size_t high = 0;
  void reset() {
    free_.clear();
    free_[0] = (size_t)1 << 60;
    high = 0;
  }
  size_t alloc(size_t bytes) {
    bytes = align_up(bytes ? bytes : 1);
    for (auto it = free_.begin(); it != free_.end(); ++it) {
      if (it->second >= bytes) {
        const size_t off = it->first, rest = it->second - bytes;
        free_.erase(it);
        if (rest) free_[off + bytes] = rest;
        if (off + bytes > high) high = off + bytes;
        return off;
      }
    }
    return (size_t)-1;  // unreachable: the last block is practically unbounded
  }
  void release(size_t off, size_t bytes) {
    bytes = align_up(bytes ? bytes : 1);
    auto it = free_.emplace(off, bytes).first;
    auto nx = std::next(it);
    if (nx != free_.end() && it->first + it->second == nx->first) {
      it->second += nx->second;
      free_.erase(nx);
    }
    if (it != free_.begin()) {
      auto pv = std::prev(it);
      if (pv->first + pv->second == it->first) {
        pv->second += it->second;
        free_.erase(it);
      }
    }
  }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

int ensure_buf(DevBuf& b, size_t bytes, hipStream_t s, bool keep_contents = false) {
  if (bytes <= b.bytes) return TLN_OK;
  size_t want = b.bytes ? b.bytes : (size_t)1 << 20;
  while (want < bytes) want += want / 2 + kAlign;
  want = align_up(want);
  void* np = nullptr;
  TLN_HIP(hipStreamSynchronize(s));
  TLN_HIP(hipMalloc(&np, want));
  if (keep_contents && b.p && b.bytes) TLN_HIP(hipMemcpy(np, b.p, b.bytes, hipMemcpyDeviceToDevice));
  if (b.p) (void)hipFree(b.p);
  b.p = np;
  b.bytes = want;
  return TLN_OK;
}

// one resolved gather-GEMM launch of the last frame (measurement: tln_program_replay_gemms)
struct GemmCall {
  int64_t M;
  int N;
  tln_gemm_src a[2];
  bool two;
  const float* w;
  int w_is_nk;
  const float* bias;
  const float* res;
  int64_t ld_res;
  int relu;
  float* out;
  int64_t ld_out;
  void* stats;
};

struct SlotRt {
  bool live = false;
  size_t off = 0, bytes = 0;   // bytes: reserved from the UPPER BOUND of the row count (same in the sizing and the
                               // launching pass, so both passes place every slot identically)
  int64_t rows = 0;            // exact row count in the launching pass
  int64_t rows_b = 0;          // upper bound
  char* ptr = nullptr;  // resolved device pointer (real pass)
};

}  // namespace

struct tln_program {
  std::vector<tln_slot> slots;
  std::vector<tln_op> ops;
  std::vector<int> last_use;  // op index after which a temporary slot can go
  int n_states = 0, n_coarse = 0;
  // sequence state
  DevBuf state_buf[TLN_MAX_STATES][2];
  int64_t state_rows[TLN_MAX_STATES] = {0};
  int state_cols[TLN_MAX_STATES] = {0};
  int state_cur[TLN_MAX_STATES] = {0};  // which of the two buffers holds the stored state
  bool state_has[TLN_MAX_STATES] = {false};
  // frame state
  tln_lattice_t* lat = nullptr;
  tln_lattice_t* levels[TLN_MAX_LEVELS] = {nullptr};
  int64_t V[TLN_MAX_LEVELS] = {0};   // exact vertex counts (coarse levels: valid once the pending fetch is finished)
  int64_t Vb[TLN_MAX_LEVELS] = {0};  // what the sizing walk plans with, known right after tln_program_begin_frame: level 0
                                     // exact; coarse levels a PREDICTION (predict_bounds) capped by the hard bound
  int64_t Vhard[TLN_MAX_LEVELS] = {0};   // the lattice's hard upper bounds (a fine vertex touches at most 4 coarse ones)
  int64_t v0_prev = 0;               // level-0 count after the previous frame of this sequence (0: none)
  double pred_scale = 1.0;           // doubled whenever a prediction was exceeded (replan)
  int64_t replans = 0;               // frames whose coarse counts exceeded the prediction
  bool exact_known = false;          // V[1..] hold the exact counts
  int split = 0;                     // ops [0, split) only touch level 0: launched before the coarse counts arrive
  int64_t N = 0;
  int dist_cols = 0;
  bool frame_open = false, frame_started = false;
  DevBuf k1;  // distributed | indices | weights
  float* d_dist = nullptr;
  int32_t* d_idx = nullptr;
  bool idx_valid = false;   // d_idx holds this frame's vertex indices (an early-return frame of a group skips them)
  float* d_w = nullptr;
  DevBuf arena;
  Arena alloc;
  std::vector<SlotRt> rt;
  bool capture = false;
  std::vector<GemmCall> calls;
  tln_options opt = tln_opt(nullptr);   // kernel-selection options of everything this program issues (tln_program_set_options)
  // stage timing (bench.py roofline_scatter): HIP events on the launch stream around K1 (every kernel of the
  // distribute), K2 (the PointNet pool) and K8 (the slice kernels of the last frame)
  bool timing = false;
  hipEvent_t tev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool tset[3] = {false, false, false};
  int timing_group = 1;   // sequences whose stage the events of this program bracket (group mode: the batch, on programs[0])
  // state of a walk that is split in two (prefix, rest)
  bool w_wrote[TLN_MAX_STATES] = {false};
  int64_t w_new_rows[TLN_MAX_STATES] = {0};
  bool w_finished = false;
  // pair mode (tln_program_run_pair): the walk stops at every gather-GEMM with its resolved call left here
  bool defer = false, has_pending = false;
  GemmCall pending{};
  // group mode: the walk also stops at every op that has a batched form (pend_kind = its TLN_OP_*, pend_op = its index)
  // with the resolved arguments left here; launch_pending issues the op of all programs of the group as ONE launch
  int pend_kind = 0, pend_op = -1;
  float* pend_pool_out = nullptr;
  tln_gn_partials_call pend_gn{};
  tln_gru_call pend_gru{};
  tln_aflow_call pend_aflow{};
  tln_slice_call pend_slice{};
  struct {
    const float* src;
    int64_t ld_src;
    float* dst;
    int64_t ld_dst, rows;
    int cols, zero_row0;
  } pend_copy{};
  int w_next = 0;  // where the next walk continues
  float* aux_out = nullptr;   // where the slice head of the NEXT run also writes log-softmax(scores) (tln_program_set_aux_out)
  // segmented run (tln_program_run_begin / _until / _end: frame-sharded multi-GPU, dist.py)
  int state_level[TLN_MAX_STATES] = {0};
  bool seg_open = false, seg_levels_done = false;
  int seg_cursor = 0, seg_early = 0, seg_out_cols = 0;
  float* seg_out = nullptr;
  int64_t seg_out_rows = 0;
};

namespace {

int64_t slot_rows(const tln_program* p, const tln_slot& s, bool bound) {
  if (s.kind == TLN_SLOT_STATE_PREV) return p->state_rows[s.state];
  if (s.rows >= 0) return (bound || (s.rows > 0 && !p->exact_known)) ? p->Vb[s.rows] : p->V[s.rows];
  if (s.rows == TLN_ROWS_POINTS) return p->N;
  if (s.rows == TLN_ROWS_POINT_ROWS) return 4 * p->N;
  return p->state_rows[TLN_ROWS_STATE - s.rows];
}

size_t slot_bytes(const tln_slot& s, int64_t rows) {
  if (s.kind == TLN_SLOT_STATS) return (size_t)tln_cdiv(rows, 32) * s.cols * 2 * sizeof(double);
  return (size_t)rows * s.cols * sizeof(float);
}

bool cond_ok(const tln_program* p, const tln_op& o) {
  if (o.cond_state < 0) return true;
  return (p->state_has[o.cond_state] ? 1 : 0) == (o.cond_has ? 1 : 0);
}

// slots an op reads / writes
template <class F>
void for_inputs(const tln_op& o, F f) {
  if (o.s0.slot >= 0) f(o.s0.slot);
  if (o.s1.slot >= 0) f(o.s1.slot);
  if (o.s0.gn_stats >= 0) f(o.s0.gn_stats);
  if (o.s1.gn_stats >= 0) f(o.s1.gn_stats);
  if (o.residual >= 0) f(o.residual);
}
template <class F>
void for_outputs(const tln_op& o, F f) {
  if (o.out >= 0) f(o.out);
  if (o.stats_out >= 0) f(o.stats_out);
}

// One walk over the op list.  WALK_DRY: only the arena bookkeeping (to size the arena); WALK_RUN: launch; WALK_REPLAY:
// everything a launching walk does to the bookkeeping (buffers resolved, states marked) but nothing is launched — how
// replan() re-establishes the state behind ops that have already run after the arena moved.
enum { WALK_RUN = 0, WALK_DRY = 1, WALK_REPLAY = 2 };
int walk(tln_program* p, int mode, int early, float* d_out, int64_t out_rows, int out_cols, hipStream_t s,
         int op_begin, int op_end, bool fresh) {
  const bool dry = mode == WALK_DRY;
  const bool nolaunch = mode != WALK_RUN;
  if (fresh) {
    p->alloc.reset();
    for (auto& r : p->rt) r = SlotRt();
    for (int st = 0; st < TLN_MAX_STATES; ++st) {
      p->w_wrote[st] = false;
      p->w_new_rows[st] = 0;
    }
    p->w_finished = false;
  }
  bool* wrote = p->w_wrote;
  int64_t* new_rows = p->w_new_rows;
  char* base = reinterpret_cast<char*>(p->arena.p);
  bool& finished = p->w_finished;

  auto tmp_alloc = [&](size_t bytes, size_t* off) {
    *off = p->alloc.alloc(bytes);
    return base ? base + *off : nullptr;
  };

  // make sure slot `id` has storage (called for op outputs)
  auto materialise = [&](int id) -> int {
    const tln_slot& sl = p->slots[id];
    SlotRt& r = p->rt[id];
    if (r.live) return TLN_OK;
    r.rows_b = slot_rows(p, sl, true);
    r.rows = dry ? r.rows_b : slot_rows(p, sl, false);
    r.bytes = slot_bytes(sl, r.rows_b);
    if (sl.kind == TLN_SLOT_STATE_NEW) {
      const int st = sl.state;
      if (!dry) {
        DevBuf& b = p->state_buf[st][1 - p->state_cur[st]];
        int rc = ensure_buf(b, r.bytes ? r.bytes : kAlign, s);
        if (rc) return rc;
        r.ptr = reinterpret_cast<char*>(b.p);
      }
      wrote[st] = true;
      new_rows[st] = r.rows;
      p->state_cols[st] = sl.cols;
    } else if (sl.kind == TLN_SLOT_OUT) {
      if (!dry) {
        TLN_REQUIRE(d_out && r.rows == out_rows && sl.cols == out_cols,
                    "program output is [%lld,%d], caller gave [%lld,%d]", (long long)r.rows, sl.cols,
                    (long long)out_rows, out_cols);
        r.ptr = reinterpret_cast<char*>(d_out);
      }
    } else if (sl.kind == TLN_SLOT_STATE_PREV) {
      TLN_REQUIRE(false, "a stored hidden state cannot be an op output");
    } else {
      r.ptr = tmp_alloc(r.bytes, &r.off);
    }
    r.live = true;
    return TLN_OK;
  };

  auto resolve_in = [&](int id) -> int {
    const tln_slot& sl = p->slots[id];
    SlotRt& r = p->rt[id];
    if (sl.kind == TLN_SLOT_STATE_PREV) {
      TLN_REQUIRE(p->state_has[sl.state], "op reads hidden state %d before it exists", sl.state);
      r.rows = r.rows_b = p->state_rows[sl.state];
      r.bytes = slot_bytes(sl, r.rows);
      r.ptr = reinterpret_cast<char*>(p->state_buf[sl.state][p->state_cur[sl.state]].p);
      r.live = true;
      return TLN_OK;
    }
    TLN_REQUIRE(r.live, "op reads slot %d before anything wrote it", id);
    return TLN_OK;
  };

  auto fptr = [&](int id) { return reinterpret_cast<float*>(p->rt[id].ptr); };

  for (int oi = op_begin; oi < op_end && !finished; ++oi) {
    const tln_op& o = p->ops[oi];
    if (!cond_ok(p, o)) continue;
    int rc = TLN_OK;
    for_inputs(o, [&](int id) {
      if (!rc) rc = resolve_in(id);
    });
    if (rc) return rc;
    if (o.kind != TLN_OP_ZERO_ROW0 && o.kind != TLN_OP_STOP_IF_EARLY) {
      for_outputs(o, [&](int id) {
        if (!rc) rc = materialise(id);
      });
      if (rc) return rc;
    }
    bool stop_here = false;
    // op-local scratch (released right after the op)
    size_t scratch_off[3] = {0, 0, 0}, scratch_bytes[3] = {0, 0, 0};
    char* scratch[3] = {nullptr, nullptr, nullptr};
    auto want_scratch = [&](int k, size_t bytes) {
      scratch_bytes[k] = bytes;
      scratch[k] = tmp_alloc(bytes, &scratch_off[k]);
    };

    switch (o.kind) {
      case TLN_OP_GEMM: {
        const tln_slot& so = p->slots[o.out];
        const int64_t M = p->rt[o.out].rows;
        if (o.s0.gn_stats >= 0) want_scratch(0, (size_t)2 * p->slots[o.s0.slot].cols * sizeof(float));
        if (nolaunch) break;
        tln_gemm_src a[2];
        const tln_op_src* os[2] = {&o.s0, &o.s1};
        for (int k = 0; k < 2; ++k) {
          if (os[k]->slot < 0) continue;
          const tln_slot& ss = p->slots[os[k]->slot];
          tln_gemm_src& g = a[k];
          g = tln_gemm_src{};
          g.d_src = fptr(os[k]->slot);
          g.src_rows = p->rt[os[k]->slot].rows;
          g.ld = ss.cols;
          g.cin = ss.cols;
          g.taps = os[k]->table == TLN_TABLE_NONE ? 1 : TLN_TAPS;
          g.pad_value = os[k]->pad_value;
          g.relu = os[k]->relu;
          if (os[k]->table != TLN_TABLE_NONE) {
            tln_lattice_t* lv = p->levels[os[k]->level];
            TLN_REQUIRE(lv, "op %d: level %d does not exist", oi, os[k]->level);
            const int32_t* tp = nullptr;
            if (os[k]->table == TLN_TABLE_NBR) rc = tln_neighbour_table(lv, &tp, s);
            else if (os[k]->table == TLN_TABLE_C2F) rc = tln_coarse_to_fine_table(lv, &tp, s);
            else rc = tln_fine_to_coarse_table(lv, &tp, s);
            if (rc) return rc;
            g.d_table = tp;
          }
          if (os[k]->gn_stats >= 0) {
            TLN_REQUIRE(k == 0, "only source 0 takes a GroupNorm prologue");
            g.d_gn_partials = p->rt[os[k]->gn_stats].ptr;
            g.d_gn_gamma = os[k]->gn_gamma;
            g.d_gn_beta = os[k]->gn_beta;
            g.gn_rows = p->rt[os[k]->slot].rows;
            g.gn_groups = os[k]->gn_groups;
            g.gn_eps = os[k]->gn_eps;
            g.d_scale = reinterpret_cast<float*>(scratch[0]);
            g.d_shift = g.d_scale + ss.cols;
          }
        }
        const float* res = o.residual >= 0 ? fptr(o.residual) : nullptr;
        const int64_t ld_res = o.residual >= 0 ? p->slots[o.residual].cols : 0;
        if (p->defer) {  // the caller launches it together with the other sequence's product
          p->pending = GemmCall{M, o.n, {a[0], a[1]}, o.s1.slot >= 0, o.w, o.w_is_nk, o.bias, res, ld_res, o.relu,
                                fptr(o.out) + o.out_col, so.cols, o.stats_out >= 0 ? p->rt[o.stats_out].ptr : nullptr};
          if (!p->pending.two) p->pending.a[1] = tln_gemm_src{};
          p->has_pending = true;
          p->pend_kind = TLN_OP_GEMM;
          p->pend_op = oi;
          stop_here = true;
          break;
        }
        {
          const tln_gemm_call call{M, o.n, &a[0], o.s1.slot >= 0 ? &a[1] : nullptr, o.w, o.w_is_nk, o.bias, res, ld_res,
                                   o.relu, fptr(o.out) + o.out_col, so.cols,
                                   o.stats_out >= 0 ? p->rt[o.stats_out].ptr : nullptr};
          rc = tln_gather_gemm_opt(&call, &p->opt, s);
        }
        if (rc) return rc;
        if (p->capture && M > 0) {
          GemmCall c{M, o.n, {a[0], a[1]}, o.s1.slot >= 0, o.w, o.w_is_nk, o.bias, res, ld_res, o.relu,
                     fptr(o.out) + o.out_col, so.cols, o.stats_out >= 0 ? p->rt[o.stats_out].ptr : nullptr};
          if (!c.two) c.a[1] = tln_gemm_src{};
          p->calls.push_back(c);
        }
        break;
      }
      case TLN_OP_GN_PARTIALS: {
        if (nolaunch) break;
        const tln_slot& ss = p->slots[o.s0.slot];
        if (p->defer && group_batches(p->opt, TLN_OP_GN_PARTIALS)) {
          p->pend_gn = tln_gn_partials_call{fptr(o.s0.slot), p->rt[o.s0.slot].rows, p->rt[o.stats_out].ptr};
          p->has_pending = true;
          p->pend_kind = TLN_OP_GN_PARTIALS;
          p->pend_op = oi;
          stop_here = true;
          break;
        }
        if (p->rt[o.s0.slot].rows > 0) {
          rc = tln_groupnorm_partials(fptr(o.s0.slot), p->rt[o.s0.slot].rows, ss.cols, p->rt[o.stats_out].ptr, s);
          if (rc) return rc;
        }
        break;
      }
      case TLN_OP_POOL: {
        if (nolaunch) break;
        const int nl = o.i[0];
        const float* w[4] = {o.p[0], o.p[1], o.p[2], o.p[3]};
        const float* b[4] = {o.p[4], o.p[5], o.p[6], o.p[7]};
        int dims[6] = {o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], 0};
        if (p->defer && group_batches(p->opt, TLN_OP_POOL)) {
          p->pend_pool_out = fptr(o.out);
          p->has_pending = true;
          p->pend_kind = TLN_OP_POOL;
          p->pend_op = oi;
          stop_here = true;
          break;
        }
        if (p->timing) TLN_HIP(hipEventRecord(p->tev[2], s));
        rc = tln_pointnet_pool(p->lat, p->d_dist, 4 * p->N, p->dist_cols, nl, w, b, dims, o.i[6], fptr(o.out), s);
        if (rc) return rc;
        if (p->timing) {
          TLN_HIP(hipEventRecord(p->tev[3], s));
          p->tset[1] = true;
        }
        break;
      }
      case TLN_OP_GRU: {
        const int64_t Vr = p->rt[o.out].rows;
        const int Cn = p->slots[o.out].cols;
        want_scratch(0, (size_t)p->rt[o.out].rows_b * 6 * Cn * sizeof(float));
        if (nolaunch) break;
        if (p->defer && !p->capture && group_batches(p->opt, TLN_OP_GRU)) {
          p->pend_gru = tln_gru_call{fptr(o.s0.slot), fptr(o.s1.slot), Vr, p->rt[o.s1.slot].rows, fptr(o.out),
                                     reinterpret_cast<float*>(scratch[0]), Vr * 6 * Cn};
          p->has_pending = true;
          p->pend_kind = TLN_OP_GRU;
          p->pend_op = oi;
          stop_here = true;
          break;
        }
        rc = tln_gru_cell_opt(fptr(o.s0.slot), fptr(o.s1.slot), Vr, p->rt[o.s1.slot].rows, Cn, o.p[0], o.p[1], o.p[2], o.p[3],
                              fptr(o.out), reinterpret_cast<float*>(scratch[0]), Vr * 6 * Cn, &p->opt, s);
        if (rc) return rc;
        if (p->capture && Vr > 0) {   // the cell's two products x @ W_ih^T, h @ W_hh^T as tln_gru_cell issues them
          float* gi = reinterpret_cast<float*>(scratch[0]);
          tln_gemm_src sx{};
          sx.d_src = fptr(o.s0.slot);
          sx.src_rows = Vr;
          sx.ld = Cn;
          sx.cin = Cn;
          sx.taps = 1;
          tln_gemm_src sh = sx;
          sh.d_src = fptr(o.s1.slot);
          sh.src_rows = p->rt[o.s1.slot].rows;
          p->calls.push_back(GemmCall{Vr, 3 * Cn, {sx, tln_gemm_src{}}, false, o.p[0], 1, o.p[2], nullptr, 0, 0, gi,
                                      3 * (int64_t)Cn, nullptr});
          p->calls.push_back(GemmCall{Vr, 3 * Cn, {sh, tln_gemm_src{}}, false, o.p[1], 1, o.p[3], nullptr, 0, 0,
                                      gi + Vr * 3 * (int64_t)Cn, 3 * (int64_t)Cn, nullptr});
        }
        break;
      }
      case TLN_OP_AFLOW: {
        const int64_t Vr = p->rt[o.out].rows;
        const int Cn = p->slots[o.out].cols;
        want_scratch(0, (size_t)p->rt[o.out].rows_b * TLN_TAPS * sizeof(float));
        want_scratch(1, (size_t)p->rt[o.out].rows_b * TLN_TAPS * sizeof(int32_t));
        if (nolaunch) break;
        const int32_t* tp = nullptr;
        TLN_REQUIRE(p->levels[o.s0.level], "op %d: level %d does not exist", oi, o.s0.level);
        rc = tln_neighbour_table(p->levels[o.s0.level], &tp, s);
        if (rc) return rc;
        if (p->defer && group_batches(p->opt, TLN_OP_AFLOW)) {
          p->pend_aflow = tln_aflow_call{fptr(o.s0.slot), fptr(o.s1.slot), Vr, p->rt[o.s1.slot].rows, tp, fptr(o.out),
                                         reinterpret_cast<float*>(scratch[0]), reinterpret_cast<int32_t*>(scratch[1])};
          p->has_pending = true;
          p->pend_kind = TLN_OP_AFLOW;
          p->pend_op = oi;
          stop_here = true;
          break;
        }
        rc = tln_aflow(fptr(o.s0.slot), fptr(o.s1.slot), Vr, p->rt[o.s1.slot].rows, Cn, tp, o.f[0], o.f[1], o.f[2],
                       o.i[0], o.bias, fptr(o.out), reinterpret_cast<float*>(scratch[0]),
                       reinterpret_cast<int32_t*>(scratch[1]), s);
        if (rc) return rc;
        break;
      }
      case TLN_OP_SLICE_GATHER: {
        if (nolaunch) break;
        TLN_REQUIRE(p->idx_valid, "op %d: the frame was begun without its vertex indices", oi);
        if (p->timing && !p->tset[2]) {
          TLN_HIP(hipEventRecord(p->tev[4], s));
          p->tset[2] = true;
        }
        rc = tln_slice_gather(fptr(o.s0.slot), p->rt[o.s0.slot].rows, p->slots[o.s0.slot].cols, p->d_idx, p->d_w, p->N,
                              fptr(o.out), s);
        if (rc) return rc;
        if (p->timing) TLN_HIP(hipEventRecord(p->tev[5], s));
        break;
      }
      case TLN_OP_SLICE: {
        if (nolaunch) break;
        TLN_REQUIRE(p->idx_valid, "op %d: the frame was begun without its vertex indices", oi);
        if (p->timing && !p->tset[2]) {
          TLN_HIP(hipEventRecord(p->tev[4], s));
          p->tset[2] = true;
        }
        rc = tln_slice(fptr(o.s0.slot), p->rt[o.s0.slot].rows, p->slots[o.s0.slot].cols, p->d_idx, p->d_w,
                       o.s1.slot >= 0 ? fptr(o.s1.slot) : nullptr, o.bias, p->N, fptr(o.out), s);
        if (rc) return rc;
        if (p->timing) TLN_HIP(hipEventRecord(p->tev[5], s));
        break;
      }
      case TLN_OP_SLICE_DEFORM: {
        if (nolaunch) break;
        TLN_REQUIRE(p->idx_valid, "op %d: the frame was begun without its vertex indices", oi);
        if (p->defer && group_batches(p->opt, TLN_OP_SLICE_DEFORM)) {
          const int ncls = p->slots[o.s1.slot].cols;
          p->pend_slice = tln_slice_call{fptr(o.s0.slot), fptr(o.s1.slot), p->rt[o.s1.slot].rows, p->d_idx, p->d_w, p->N,
                                         fptr(o.out), ncls <= 64 ? p->aux_out : nullptr};
          p->aux_out = nullptr;
          p->has_pending = true;
          p->pend_kind = TLN_OP_SLICE_DEFORM;
          p->pend_op = oi;
          stop_here = true;
          break;
        }
        if (p->timing && !p->tset[2]) {
          TLN_HIP(hipEventRecord(p->tev[4], s));
          p->tset[2] = true;
        }
        rc = tln_slice_deform_ls(fptr(o.s0.slot), p->slots[o.s0.slot].cols, fptr(o.s1.slot), p->rt[o.s1.slot].rows,
                                 p->slots[o.s1.slot].cols, p->d_idx, p->d_w, o.p[0], o.p[1], o.p[2], o.bias, p->N,
                                 fptr(o.out), p->slots[o.s1.slot].cols <= 64 ? p->aux_out : nullptr, s);
        if (rc) return rc;
        p->aux_out = nullptr;
        if (p->timing) TLN_HIP(hipEventRecord(p->tev[5], s));
        break;
      }
      case TLN_OP_LSTM_GATES: {
        if (nolaunch) break;
        rc = tln_lstm_gates(fptr(o.s0.slot), p->rt[o.out].rows, p->slots[o.out].cols, fptr(o.out), s);
        if (rc) return rc;
        break;
      }
      case TLN_OP_TEMPORAL_MAX: {
        if (nolaunch) break;
        rc = tln_temporal_max(fptr(o.s0.slot), fptr(o.s1.slot), p->rt[o.out].rows, p->rt[o.s1.slot].rows,
                              p->slots[o.out].cols, o.f[0], fptr(o.out), s);
        if (rc) return rc;
        break;
      }
      case TLN_OP_CGA_GATE: {
        if (nolaunch) break;
        const int64_t Vr = p->rt[o.out].rows;
        const int Cn = p->slots[o.out].cols;
        TLN_REQUIRE(o.i[0] >= 0 && o.i[0] < p->n_states, "op %d: bad state id", oi);
        rc = tln_cga_gate(fptr(o.s0.slot), fptr(o.s1.slot), Vr, p->state_rows[o.i[0]], Cn, 1.0f / (float)(Vr + Cn),
                          fptr(o.out), s);
        if (rc) return rc;
        break;
      }
      case TLN_OP_FILL_EMPTY: {
        if (nolaunch) break;
        rc = tln_fill_empty_rows(fptr(o.s0.slot), p->rt[o.out].rows, p->slots[o.out].cols, o.i[0], o.f[0], fptr(o.out),
                                 s);
        if (rc) return rc;
        break;
      }
      case TLN_OP_COPY: {
        if (nolaunch) break;
        const tln_slot& ss = p->slots[o.s0.slot];
        const tln_slot& so = p->slots[o.out];
        const int64_t rows = p->rt[o.s0.slot].rows;
        TLN_REQUIRE(rows == p->rt[o.out].rows && o.out_col + ss.cols <= so.cols, "op %d: copy shapes", oi);
        if (rows > 0) {
          const float* sp = fptr(o.s0.slot);
          float* dp = fptr(o.out) + o.out_col;
          const bool vec = ss.cols % 4 == 0 && so.cols % 4 == 0 && o.out_col % 4 == 0 &&
                           (reinterpret_cast<uintptr_t>(sp) & 15) == 0 && (reinterpret_cast<uintptr_t>(dp) & 15) == 0;
          if (vec && p->defer && group_batches(p->opt, TLN_OP_COPY)) {
            p->pend_copy.src = sp;
            p->pend_copy.ld_src = ss.cols;
            p->pend_copy.dst = dp;
            p->pend_copy.ld_dst = so.cols;
            p->pend_copy.rows = rows;
            p->pend_copy.cols = ss.cols;
            p->pend_copy.zero_row0 = o.i[0];
            p->has_pending = true;
            p->pend_kind = TLN_OP_COPY;
            p->pend_op = oi;
            stop_here = true;
            break;
          }
          if (vec) {
            const int64_t total = rows * (ss.cols / 4);
            hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)tln_cdiv(total, 256)), dim3(256), 0, s, sp, (int64_t)ss.cols, dp,
                               (int64_t)so.cols, rows, ss.cols, o.i[0]);
            TLN_LAUNCH_CHECK();
          } else {
            TLN_HIP(hipMemcpy2DAsync(dp, (size_t)so.cols * sizeof(float), sp, (size_t)ss.cols * sizeof(float),
                                     (size_t)ss.cols * sizeof(float), (size_t)rows, hipMemcpyDeviceToDevice, s));
            if (o.i[0]) TLN_HIP(hipMemsetAsync(dp, 0, (size_t)ss.cols * sizeof(float), s));
          }
        }
        break;
      }
      case TLN_OP_ZERO_ROW0: {
        TLN_REQUIRE(p->rt[o.out].live, "op %d: zero-row on an unwritten slot", oi);
        if (nolaunch) break;
        if (p->rt[o.out].rows > 0)
          TLN_HIP(hipMemsetAsync(fptr(o.out), 0, (size_t)p->slots[o.out].cols * sizeof(float), s));
        break;
      }
      case TLN_OP_STOP_IF_EARLY: {
        if (!early) break;
        finished = true;
        if (nolaunch) break;
        const tln_slot& ss = p->slots[o.s0.slot];
        const int64_t rows = p->rt[o.s0.slot].rows;
        // (d_out NULL: the caller does not want the value — the reference returns the hidden-state tensor itself there,
        //  models.py:430, and its loops drop it; here it would be a copy out of the program's state buffer)
        TLN_REQUIRE(!d_out || (rows == out_rows && ss.cols == out_cols),
                    "early-return value is [%lld,%d], caller gave [%lld,%d]", (long long)rows, ss.cols,
                    (long long)out_rows, out_cols);
        if (rows > 0 && d_out)
          TLN_HIP(hipMemcpyAsync(d_out, fptr(o.s0.slot), (size_t)rows * ss.cols * sizeof(float),
                                 hipMemcpyDeviceToDevice, s));
        break;
      }
      default:
        TLN_REQUIRE(false, "op %d: unknown kind %d", oi, o.kind);
    }
    for (int k = 0; k < 3; ++k)
      if (scratch_bytes[k]) p->alloc.release(scratch_off[k], scratch_bytes[k]);
    // temporaries nobody reads any more
    auto drop = [&](int id) {
      SlotRt& r = p->rt[id];
      if (r.live && p->last_use[id] <= oi && p->slots[id].kind != TLN_SLOT_STATE_NEW &&
          p->slots[id].kind != TLN_SLOT_STATE_PREV && p->slots[id].kind != TLN_SLOT_OUT) {
        p->alloc.release(r.off, r.bytes);
        r.live = false;
      }
    };
    for_inputs(o, drop);
    for_outputs(o, drop);
    if (stop_here) {
      p->w_next = oi + 1;
      return TLN_OK;
    }
  }
  p->w_next = op_end;
  return TLN_OK;
}

// Planning sizes of the coarse levels.  The lattice only knows HARD bounds before its counters arrive (level i: old count
// + 4^i x the new level-0 vertices; the second coarse level of a first frame: 16 x V0, capped by the capacity) and the
// arena, the hidden-state buffers and every coarse temporary used to be reserved for them: ~1 GB of arena and 0.5 GB of
// state buffers per resident sequence for a working set of 0.2 GB (round 3's "2 GB per sequence").  The walk plans with a
// PREDICTION instead — old count + max(2048, r_i x new level-0 vertices), r = 1, 1/2, 1/4 ... (measured: 0.3-0.45 and
// 0.1-0.15 on lidar lattices from sigma 0.07 to 1.0), times pred_scale — and replan() repairs the rare frame whose exact
// counts exceed it.
void predict_bounds(tln_program* p, const int64_t* hard) {
  const int64_t v0 = hard[0];
  const int64_t dv0 = (p->v0_prev > 0 && p->v0_prev <= v0) ? v0 - p->v0_prev : v0;
  double r = 1.0;
  p->Vhard[0] = p->Vb[0] = v0;
  for (int i = 1; i <= p->n_coarse; ++i) {
    p->Vhard[i] = hard[i];
    const int64_t old = p->levels[i] ? tln_lattice_nr_vertices(p->levels[i]) : 0;   // (the count before this frame)
    const double grow = r * p->pred_scale * (double)dv0;
    int64_t pred = (old > 0 ? old : 0) + (int64_t)(grow < 2048.0 ? 2048.0 : grow);
    p->Vb[i] = pred < hard[i] ? pred : hard[i];
    r *= 0.5;
  }
}

// exact coarse counts against the hard bounds (an error) and the prediction (true: replan needed)
int check_counts(tln_program* p, bool* exceeded) {
  *exceeded = false;
  for (int i = 1; i <= p->n_coarse; ++i) {
    p->V[i] = tln_lattice_nr_vertices(p->levels[i]);
    TLN_REQUIRE(p->V[i] <= p->Vhard[i], "level %d has %lld vertices, more than the bound %lld", i, (long long)p->V[i],
                (long long)p->Vhard[i]);
    if (p->V[i] > p->Vb[i]) *exceeded = true;
  }
  p->exact_known = true;
  return TLN_OK;
}

// The exact coarse counts exceed what the frame was planned with; ops [0, upto) have run.  Plan again with the exact
// counts (the placements of everything those ops wrote are the same: they were decided before any coarse slot), grow the
// arena with its contents kept, and re-establish the walk state behind op `upto` without launching anything.
int replan(tln_program* p, int upto, int early, float* d_out, int64_t out_rows, int out_cols, hipStream_t s) {
  for (int i = 1; i <= p->n_coarse; ++i) p->Vb[i] = p->V[i];
  p->pred_scale *= 2.0;
  ++p->replans;
  const int n_ops = (int)p->ops.size();
  int rc = walk(p, WALK_DRY, early, nullptr, out_rows, out_cols, s, 0, n_ops, true);
  if (rc) return rc;
  rc = ensure_buf(p->arena, p->alloc.high + kAlign, s, true);
  if (rc) return rc;
  return walk(p, WALK_REPLAY, early, d_out, out_rows, out_cols, s, 0, upto, true);
}

// the frame that wrote a hidden state is over: the new buffer becomes the stored one
void commit_states(tln_program* p) {
  p->v0_prev = p->V[0];
  for (int st = 0; st < p->n_states; ++st)
    if (p->w_wrote[st]) {
      p->state_cur[st] = 1 - p->state_cur[st];
      p->state_rows[st] = p->w_new_rows[st];
      p->state_has[st] = true;
    }
}

}  // namespace

extern "C" int tln_program_create(tln_program_t** out, const tln_slot* slots, int n_slots, const tln_op* ops, int n_ops,
                                  int n_states, int nr_coarse_levels) {
  TLN_REQUIRE(out && slots && ops && n_slots > 0 && n_ops > 0, "null argument");
  TLN_REQUIRE(n_states >= 0 && n_states <= TLN_MAX_STATES, "at most %d hidden states", TLN_MAX_STATES);
  TLN_REQUIRE(nr_coarse_levels >= 0 && nr_coarse_levels < TLN_MAX_LEVELS, "at most %d levels", TLN_MAX_LEVELS);
  tln_program* p = new tln_program();
  p->slots.assign(slots, slots + n_slots);
  p->ops.assign(ops, ops + n_ops);
  p->n_states = n_states;
  p->n_coarse = nr_coarse_levels;
  p->rt.resize(n_slots);
  p->last_use.assign(n_slots, -1);
  auto fail = [&](const char* what, int oi) {
    tln_set_error("program op %d: %s", oi, what);
    delete p;
    return TLN_E_INVALID;
  };
  for (int i = 0; i < n_slots; ++i) {
    const tln_slot& s = p->slots[i];
    const bool rows_ok = (s.rows >= 0 && s.rows <= nr_coarse_levels) || s.rows == TLN_ROWS_POINTS ||
                         s.rows == TLN_ROWS_POINT_ROWS || (s.rows <= TLN_ROWS_STATE && TLN_ROWS_STATE - s.rows < n_states);
    if (!rows_ok || s.cols <= 0) return fail("bad slot rows / cols", -1 - i);
    if ((s.kind == TLN_SLOT_STATE_NEW || s.kind == TLN_SLOT_STATE_PREV) && (s.state < 0 || s.state >= n_states))
      return fail("bad state id", -1 - i);
  }
  for (int oi = 0; oi < n_ops; ++oi) {
    const tln_op& o = p->ops[oi];
    bool ok = true;
    auto chk = [&](int id) {
      if (id < 0 || id >= n_slots) ok = false;
      else p->last_use[id] = oi;
    };
    for_inputs(o, chk);
    for_outputs(o, chk);
    if (!ok) return fail("slot id out of range", oi);
    if (o.cond_state >= n_states) return fail("bad condition state", oi);
    if (o.kind == TLN_OP_GEMM && (!o.w || o.n <= 0 || o.s0.slot < 0)) return fail("incomplete GEMM", oi);
  }
  // ops [0, split) touch level 0 only (slots, hidden states and tables): they can be launched before the coarse levels'
  // vertex counts are known on the host
  {
    int* state_level = p->state_level;
    for (int st = 0; st < TLN_MAX_STATES; ++st) state_level[st] = 0;
    for (const tln_slot& sl : p->slots)
      if (sl.kind == TLN_SLOT_STATE_NEW && sl.rows >= 0) state_level[sl.state] = sl.rows;
    auto slot_coarse = [&](int id) {
      if (id < 0) return false;
      const tln_slot& sl = p->slots[id];
      if (sl.kind == TLN_SLOT_STATE_PREV || sl.kind == TLN_SLOT_STATE_NEW) return state_level[sl.state] > 0;
      if (sl.rows > 0) return true;
      if (sl.rows <= TLN_ROWS_STATE) return state_level[TLN_ROWS_STATE - sl.rows] > 0;
      return false;
    };
    p->split = n_ops;
    for (int oi = 0; oi < n_ops; ++oi) {
      const tln_op& o = p->ops[oi];
      bool coarse = slot_coarse(o.out) || slot_coarse(o.stats_out) || slot_coarse(o.s0.slot) || slot_coarse(o.s1.slot) ||
                    slot_coarse(o.s0.gn_stats) || slot_coarse(o.residual);
      if (o.s0.slot >= 0 && o.s0.table != TLN_TABLE_NONE && o.s0.level > 0) coarse = true;
      if (o.s1.slot >= 0 && o.s1.table != TLN_TABLE_NONE && o.s1.level > 0) coarse = true;
      if (o.kind == TLN_OP_AFLOW && o.s0.level > 0) coarse = true;
      if (o.kind == TLN_OP_CGA_GATE && state_level[o.i[0]] > 0) coarse = true;
      if (coarse) {
        p->split = oi;
        break;
      }
    }
  }
  *out = p;
  return TLN_OK;
}

extern "C" int tln_program_destroy(tln_program_t* p) {
  if (!p) return TLN_OK;
  (void)hipDeviceSynchronize();
  for (int st = 0; st < TLN_MAX_STATES; ++st)
    for (int k = 0; k < 2; ++k)
      if (p->state_buf[st][k].p) (void)hipFree(p->state_buf[st][k].p);
  if (p->k1.p) (void)hipFree(p->k1.p);
  if (p->arena.p) (void)hipFree(p->arena.p);
  for (int i = 0; i < 6; ++i)
    if (p->tev[i]) (void)hipEventDestroy(p->tev[i]);
  delete p;
  return TLN_OK;
}

extern "C" int tln_program_reset(tln_program_t* p) {
  TLN_REQUIRE(p, "null program");
  for (int st = 0; st < TLN_MAX_STATES; ++st) {
    p->state_has[st] = false;
    p->state_rows[st] = 0;
  }
  p->frame_open = false;
  p->v0_prev = 0;
  return TLN_OK;
}

// the frame's K1 outputs (distributed rows if wanted | indices | weights) placed in the program's buffer
static int frame_buffers(tln_program* p, int64_t n, int val_dim, hipStream_t s) {
  const int cols = 3 + val_dim + 1;
  // the [4N, 5] `distributed` rows are not materialised when the pool can take the frame's rows from the lattice's
  // vertex bins (one value channel: every supported PointNet shape); the pool then gets d_distributed = NULL
  const bool want_dist = val_dim != 1;
  const size_t dist_b = want_dist ? align_up((size_t)4 * n * cols * sizeof(float)) : 0;
  const size_t idx_b = align_up((size_t)4 * n * sizeof(int32_t));
  int rc = ensure_buf(p->k1, dist_b + 2 * idx_b, s);
  if (rc) return rc;
  char* b = reinterpret_cast<char*>(p->k1.p);
  p->d_dist = want_dist ? reinterpret_cast<float*>(b) : nullptr;
  p->d_idx = reinterpret_cast<int32_t*>(b + dist_b);
  p->d_w = reinterpret_cast<float*>(b + dist_b + idx_b);
  return TLN_OK;
}

static void frame_started(tln_program* p, tln_lattice_t* l, int64_t n, int val_dim) {
  p->lat = l;
  p->N = n;
  p->dist_cols = 3 + val_dim + 1;
  p->frame_open = false;
  p->frame_started = true;
}

// first half of tln_program_begin_frame: K1 up to the point where the vertex counters start their way to the host
extern "C" int tln_program_begin_frame_start(tln_program_t* p, tln_lattice_t* l, const float* d_positions,
                                             const float* d_values, int64_t n, int val_dim, int reset_hashmap,
                                             int subtract_mean, void* stream_) {
  TLN_REQUIRE(p && l && d_positions && n > 0 && val_dim >= 0, "bad frame arguments");
  hipStream_t s = (hipStream_t)stream_;
  int rc = frame_buffers(p, n, val_dim, s);
  if (rc) return rc;
  if (reset_hashmap) {
    rc = tln_lattice_clear(l, s);
    if (rc) return rc;
    p->v0_prev = 0;
  }
  p->timing_group = 1;
  if (p->timing) {
    p->tset[0] = p->tset[1] = p->tset[2] = false;
    TLN_HIP(hipEventRecord(p->tev[0], s));
  }
  p->idx_valid = true;
  rc = tln_distribute_begin(l, d_positions, d_values, n, val_dim, subtract_mean, p->d_dist, p->d_idx, p->d_w, s);
  if (rc) return rc;
  if (p->timing) {   // every kernel of K1 is enqueued by now (the second half only waits for the counters)
    TLN_HIP(hipEventRecord(p->tev[1], s));
    p->tset[0] = true;
  }
  frame_started(p, l, n, val_dim);
  return TLN_OK;
}

// the frames of `count` lock-stepped sequences begun together: ONE batch of K1 launches for all of them
// (tln_distribute_begin_multi: blockIdx.y = sequence), one wait for the vertex counters, then every program's second
// half.  v_out: [count][TLN_MAX_LEVELS].  With stage timing on, programs[0] holds the events around the whole batch.
extern "C" int tln_program_begin_frame_group(tln_program_t* const* pp, tln_lattice_t* const* ll,
                                             const float* const* d_positions, const float* const* d_values,
                                             const int64_t* n, int count, int val_dim, int reset_hashmap, int subtract_mean,
                                             int need_indices, int64_t* v_out, void* stream_) {
  TLN_REQUIRE(pp && ll && d_positions && n && v_out && count >= 1 && count <= 8 && val_dim >= 0, "bad frame group");
  hipStream_t s = (hipStream_t)stream_;
  if (!group_batches(pp[0]->opt, 0)) {   // every sequence by itself: all first halves, then all second halves
    for (int k = 0; k < count; ++k) {
      int rc = tln_program_begin_frame_start(pp[k], ll[k], d_positions[k], d_values ? d_values[k] : nullptr, n[k], val_dim,
                                             reset_hashmap, subtract_mean, s);
      if (rc) return rc;
    }
    (void)need_indices;
    for (int k = 0; k < count; ++k) {
      int rc = tln_program_begin_frame_finish(pp[k], v_out + (size_t)k * TLN_MAX_LEVELS, s);
      if (rc) return rc;
    }
    return TLN_OK;
  }
  tln_distribute_call calls[8];
  for (int k = 0; k < count; ++k) {
    TLN_REQUIRE(pp[k] && ll[k] && d_positions[k] && n[k] > 0, "bad frame arguments (sequence %d)", k);
    int rc = frame_buffers(pp[k], n[k], val_dim, s);
    if (rc) return rc;
  }
  if (reset_hashmap) {
    int rc = tln_lattice_clear_multi(ll, count, s);
    if (rc) return rc;
    for (int k = 0; k < count; ++k) pp[k]->v0_prev = 0;
  }
  tln_program* p0 = pp[0];
  for (int k = 0; k < count; ++k) {
    pp[k]->tset[0] = pp[k]->tset[1] = pp[k]->tset[2] = false;
    pp[k]->timing_group = count;
    // the per-row vertex indices are read by the slice ops only: a frame that returns early does not need them (and
    // k_bk_place saves a scattered 4-byte store per row); the [4N, .] rows, if the program wants them, need them
    const bool want_idx = need_indices != 0 || pp[k]->d_dist != nullptr;
    pp[k]->idx_valid = want_idx;
    calls[k] = tln_distribute_call{ll[k], d_positions[k], d_values ? d_values[k] : nullptr, n[k], val_dim, subtract_mean,
                                   pp[k]->d_dist, want_idx ? pp[k]->d_idx : nullptr, pp[k]->d_w};
  }
  if (p0->timing) TLN_HIP(hipEventRecord(p0->tev[0], s));
  int rc = tln_distribute_begin_multi(calls, count, s);
  if (rc) return rc;
  if (p0->timing) {
    TLN_HIP(hipEventRecord(p0->tev[1], s));
    p0->tset[0] = true;
  }
  for (int k = 0; k < count; ++k) frame_started(pp[k], ll[k], n[k], val_dim);
  // second halves: every lattice's vertex counters (one wait: the batch shares an event), then the coarse levels of all
  // of them extended by one batch of launches
  for (int k = 0; k < count; ++k) {
    pp[k]->frame_started = false;
    rc = tln_distribute_finish(ll[k], s);
    if (rc) return rc;
    TLN_REQUIRE(pp[k]->n_coarse == p0->n_coarse, "the programs of a group differ in their level count");
  }
  int64_t vb[8 * TLN_MAX_LEVELS] = {0};
  rc = tln_lattice_prepare_levels_begin_multi(ll, count, p0->n_coarse, vb, s);
  if (rc) return rc;
  for (int k = 0; k < count; ++k) {
    tln_program* p = pp[k];
    p->exact_known = p->n_coarse == 0;
    tln_lattice_t* lv = ll[k];
    for (int i = 0; i <= p->n_coarse; ++i) {
      TLN_REQUIRE(lv, "level %d does not exist", i);
      p->levels[i] = lv;
      p->V[i] = i == 0 ? vb[(size_t)k * TLN_MAX_LEVELS] : -1;
      v_out[(size_t)k * TLN_MAX_LEVELS + i] = vb[(size_t)k * TLN_MAX_LEVELS + i];
      lv = tln_lattice_coarse_level(lv);
    }
    predict_bounds(p, vb + (size_t)k * TLN_MAX_LEVELS);
    p->frame_open = true;
  }
  return TLN_OK;
}

// second half: waits for this frame's vertex counters only, then the CSR, the means and the coarse levels
extern "C" int tln_program_begin_frame_finish(tln_program_t* p, int64_t* v_out, void* stream_) {
  TLN_REQUIRE(p && p->frame_started && v_out, "tln_program_begin_frame_finish without _start");
  hipStream_t s = (hipStream_t)stream_;
  p->frame_started = false;
  tln_lattice_t* l = p->lat;
  int rc = tln_distribute_finish(l, s);
  if (rc) return rc;
  int64_t vb[TLN_MAX_LEVELS] = {0};
  rc = tln_lattice_prepare_levels_begin(l, p->n_coarse, vb, s);   // coarse counts stay in flight until tln_program_run
  if (rc) return rc;
  p->exact_known = p->n_coarse == 0;
  tln_lattice_t* lv = l;
  for (int i = 0; i <= p->n_coarse; ++i) {
    TLN_REQUIRE(lv, "level %d does not exist", i);
    p->levels[i] = lv;
    p->V[i] = i == 0 ? vb[0] : -1;
    v_out[i] = vb[i];   // level 0 exact; coarse levels: upper bounds (the exact counts follow in tln_program_run)
    lv = tln_lattice_coarse_level(lv);
  }
  predict_bounds(p, vb);
  p->frame_open = true;
  return TLN_OK;
}

extern "C" int tln_program_begin_frame(tln_program_t* p, tln_lattice_t* l, const float* d_positions,
                                       const float* d_values, int64_t n, int val_dim, int reset_hashmap,
                                       int subtract_mean, int64_t* v_out, void* stream_) {
  TLN_REQUIRE(v_out, "bad frame arguments");
  int rc = tln_program_begin_frame_start(p, l, d_positions, d_values, n, val_dim, reset_hashmap, subtract_mean, stream_);
  return rc ? rc : tln_program_begin_frame_finish(p, v_out, stream_);
}

extern "C" int tln_program_run(tln_program_t* p, int early, float* d_out, int64_t out_rows, int out_cols,
                               void* stream_) {
  TLN_REQUIRE(p && p->frame_open, "tln_program_run without tln_program_begin_frame");
  hipStream_t s = (hipStream_t)stream_;
  const int n_ops = (int)p->ops.size();
  // sizing pass over the whole op list with the upper bounds of the coarse vertex counts
  int rc = walk(p, WALK_DRY, early, nullptr, out_rows, out_cols, s, 0, n_ops, true);
  if (rc) return rc;
  rc = ensure_buf(p->arena, p->alloc.high + kAlign, s);
  if (rc) return rc;
  p->calls.clear();
  // level-0 prefix: on the GPU while the coarse counts are still travelling to the host
  rc = walk(p, WALK_RUN, early, d_out, out_rows, out_cols, s, 0, p->split, true);
  // the exact coarse counts (waits for the fetch only) and the coarse tables
  int rc2 = tln_lattice_prepare_levels_finish(p->lat, s);
  if (rc == TLN_OK) rc = rc2;
  if (rc == TLN_OK) {
    bool exceeded = false;
    rc = check_counts(p, &exceeded);
    if (rc == TLN_OK && exceeded && !p->w_finished) rc = replan(p, p->split, early, d_out, out_rows, out_cols, s);
    if (rc == TLN_OK && !p->w_finished) rc = walk(p, WALK_RUN, early, d_out, out_rows, out_cols, s, p->split, n_ops, false);
  }
  if (rc == TLN_OK) commit_states(p);
  p->frame_open = false;
  return rc;
}

// ---- the frame in segments (frame-sharded multi-GPU: a hidden state arrives from the rank that owns the previous
// frame right before the first op that reads it, and leaves for the next rank right after the last op that writes it;
// temporal_latticenet_amd/dist.py).  Same walk as tln_program_run, cut at caller-chosen op indices.
extern "C" int tln_program_nr_ops(const tln_program_t* p) { return p ? (int)p->ops.size() : -1; }

// op range of hidden state `id`: first op that reads the stored state, last op that writes the new one (-1: none),
// and the lattice level the state lives on
extern "C" int tln_program_state_ops(const tln_program_t* p, int id, int* first_read_op, int* last_write_op, int* level) {
  TLN_REQUIRE(p && id >= 0 && id < p->n_states, "bad state id");
  int fr = -1, lw = -1;
  for (int oi = 0; oi < (int)p->ops.size(); ++oi) {
    const tln_op& o = p->ops[oi];
    bool reads = o.cond_state == id;     // an op conditional on the state is decided when the walk reaches it
    for_inputs(o, [&](int sid) {
      const tln_slot& sl = p->slots[sid];
      if (sl.kind == TLN_SLOT_STATE_PREV && sl.state == id) reads = true;
      if (sl.rows <= TLN_ROWS_STATE && TLN_ROWS_STATE - sl.rows == id) reads = true;
    });
    bool writes = false;
    for_outputs(o, [&](int sid) {
      const tln_slot& sl = p->slots[sid];
      if (sl.kind == TLN_SLOT_STATE_NEW && sl.state == id) writes = true;
      if (sl.rows <= TLN_ROWS_STATE && TLN_ROWS_STATE - sl.rows == id) reads = true;
    });
    if (o.kind == TLN_OP_CGA_GATE && o.i[0] == id) reads = true;
    if (reads && fr < 0) fr = oi;
    if (writes) lw = oi;
  }
  if (first_read_op) *first_read_op = fr;
  if (last_write_op) *last_write_op = lw;
  if (level) *level = p->state_level[id];
  return TLN_OK;
}

// announces a stored hidden state of `rows` rows whose contents follow later (tln_program_state_set with the same row
// count): the sizing walk of tln_program_run_begin must see every state the frame will read
extern "C" int tln_program_state_expect(tln_program_t* p, int id, int64_t rows, void* stream_) {
  TLN_REQUIRE(p && id >= 0 && id < p->n_states && rows >= 0, "bad state id");
  int cols = p->state_cols[id];
  if (cols == 0)
    for (const tln_slot& s : p->slots)
      if (s.kind == TLN_SLOT_STATE_NEW && s.state == id) cols = s.cols;
  TLN_REQUIRE(cols > 0, "hidden state %d has no width", id);
  DevBuf& b = p->state_buf[id][p->state_cur[id]];
  const size_t bytes = (size_t)rows * cols * sizeof(float);
  int rc = ensure_buf(b, bytes ? bytes : kAlign, (hipStream_t)stream_);
  if (rc) return rc;
  p->state_rows[id] = rows;
  p->state_cols[id] = cols;
  p->state_has[id] = true;
  return TLN_OK;
}

static int seg_walk_to(tln_program* p, int op_end, hipStream_t s) {
  const int n_ops = (int)p->ops.size();
  if (op_end > n_ops) op_end = n_ops;
  int rc = TLN_OK;
  if (p->seg_cursor < p->split && op_end > p->seg_cursor) {
    const int e = op_end < p->split ? op_end : p->split;
    rc = walk(p, WALK_RUN, p->seg_early, p->seg_out, p->seg_out_rows, p->seg_out_cols, s, p->seg_cursor, e, false);
    if (rc) return rc;
    p->seg_cursor = e;
  }
  if (op_end > p->split && !p->seg_levels_done) {
    rc = tln_lattice_prepare_levels_finish(p->lat, s);
    if (rc) return rc;
    bool exceeded = false;
    rc = check_counts(p, &exceeded);
    if (rc) return rc;
    if (exceeded && !p->w_finished) {
      rc = replan(p, p->seg_cursor, p->seg_early, p->seg_out, p->seg_out_rows, p->seg_out_cols, s);
      if (rc) return rc;
    }
    p->seg_levels_done = true;
  }
  if (op_end > p->seg_cursor && !p->w_finished) {
    rc = walk(p, WALK_RUN, p->seg_early, p->seg_out, p->seg_out_rows, p->seg_out_cols, s, p->seg_cursor, op_end, false);
    if (rc) return rc;
  }
  if (op_end > p->seg_cursor) p->seg_cursor = op_end;
  return TLN_OK;
}

extern "C" int tln_program_run_begin(tln_program_t* p, int early, float* d_out, int64_t out_rows, int out_cols,
                                     void* stream_) {
  TLN_REQUIRE(p && p->frame_open && !p->seg_open, "tln_program_run_begin without an open frame");
  hipStream_t s = (hipStream_t)stream_;
  const int n_ops = (int)p->ops.size();
  int rc = walk(p, WALK_DRY, early, nullptr, out_rows, out_cols, s, 0, n_ops, true);   // sizing (resets the walk state)
  if (rc) return rc;
  rc = ensure_buf(p->arena, p->alloc.high + kAlign, s);
  if (rc) return rc;
  p->calls.clear();
  // the launching walk starts from a fresh allocator too: an empty range with fresh = true
  rc = walk(p, WALK_RUN, early, d_out, out_rows, out_cols, s, 0, 0, true);
  if (rc) return rc;
  p->seg_open = true;
  p->seg_levels_done = p->n_coarse == 0;
  p->seg_cursor = 0;
  p->seg_early = early;
  p->seg_out = d_out;
  p->seg_out_rows = out_rows;
  p->seg_out_cols = out_cols;
  return TLN_OK;
}

extern "C" int tln_program_run_until(tln_program_t* p, int op_end, void* stream_) {
  TLN_REQUIRE(p && p->seg_open && op_end >= 0, "tln_program_run_until without tln_program_run_begin");
  // a cut behind the cursor means its state would be set AFTER the ops that read it have run (on a buffer that
  // tln_program_state_expect only sized): refuse instead of returning wrong logits
  TLN_REQUIRE(op_end >= p->seg_cursor, "tln_program_run_until(%d): ops up to %d have already run", op_end, p->seg_cursor);
  return seg_walk_to(p, op_end, (hipStream_t)stream_);
}

extern "C" int tln_program_run_end(tln_program_t* p, void* stream_) {
  TLN_REQUIRE(p && p->seg_open, "tln_program_run_end without tln_program_run_begin");
  int rc = seg_walk_to(p, (int)p->ops.size(), (hipStream_t)stream_);
  if (rc == TLN_OK) commit_states(p);
  p->seg_open = false;
  p->frame_open = false;
  return rc;
}

// the hidden state this frame has written so far (before the frame ends and it becomes the stored one)
extern "C" int tln_program_state_new_info(const tln_program_t* p, int id, int64_t* rows, int* cols, int* written) {
  TLN_REQUIRE(p && id >= 0 && id < p->n_states, "bad state id");
  if (rows) *rows = p->w_wrote[id] ? p->w_new_rows[id] : 0;
  if (cols) *cols = p->state_cols[id];
  if (written) *written = p->w_wrote[id] ? 1 : 0;
  return TLN_OK;
}

extern "C" int tln_program_state_get_new(tln_program_t* p, int id, float* d_out, void* stream_) {
  TLN_REQUIRE(p && id >= 0 && id < p->n_states && d_out, "bad state id");
  TLN_REQUIRE(p->w_wrote[id], "hidden state %d has not been written by this frame yet", id);
  const size_t bytes = (size_t)p->w_new_rows[id] * p->state_cols[id] * sizeof(float);
  if (bytes)
    TLN_HIP(hipMemcpyAsync(d_out, p->state_buf[id][1 - p->state_cur[id]].p, bytes, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream_));
  return TLN_OK;
}

// ---- group mode: 2..8 sequences stepped in lock-step on one stream, their gather-GEMM ops sharing launches --------
namespace {
constexpr int kMaxGroup = 8;

int launch_pending_gemms(tln_program* const* pp, int n, hipStream_t s) {
  tln_gemm_call calls[kMaxGroup];
  int m = 0;
  for (int k = 0; k < n; ++k)
    if (pp[k]->has_pending) {
      const GemmCall& c = pp[k]->pending;
      calls[m++] = tln_gemm_call{c.M, c.N, &c.a[0], c.two ? &c.a[1] : nullptr, c.w, c.w_is_nk, c.bias, c.res, c.ld_res,
                                 c.relu, c.out, c.ld_out, c.stats};
    }
  int rc = m ? tln_gather_gemm_multi_opt(calls, m, &pp[0]->opt, s) : TLN_OK;
  for (int k = 0; k < n; ++k) {
    tln_program* p = pp[k];
    if (p->has_pending && p->capture && p->pending.M > 0) p->calls.push_back(p->pending);
  }
  return rc;
}

// The op every program of the group stopped at, issued for all of them together: the products through
// tln_gather_gemm_multi, every other kind through its batched entry point (one launch, blockIdx.y / .z = sequence).  The
// batched forms share the op's parameters (weights, biases): programs compiled from different weights, or stopped at
// different ops, are served one by one with the same calls.
int launch_pending(tln_program* const* pp, int n, hipStream_t s) {
  tln_program* q[kMaxGroup];
  int m = 0;
  for (int k = 0; k < n; ++k)
    if (pp[k]->has_pending) q[m++] = pp[k];
  if (m == 0) return TLN_OK;
  bool same = true;
  for (int k = 1; k < m; ++k) same = same && q[k]->pend_kind == q[0]->pend_kind && q[k]->pend_op == q[0]->pend_op;
  int rc = TLN_OK;
  // [b, e) = the programs served by one call
  for (int b = 0; b < m && !rc;) {
    int e = b + 1;
    const tln_op& o = q[b]->ops[q[b]->pend_op];
    auto same_params = [&](const tln_op& x) {
      bool eq = x.bias == o.bias && x.w == o.w;
      for (int i = 0; i < 8; ++i) eq = eq && x.p[i] == o.p[i] && x.i[i] == o.i[i];
      for (int i = 0; i < 4; ++i) eq = eq && x.f[i] == o.f[i];
      return eq;
    };
    if (same)
      while (e < m && (q[b]->pend_kind == TLN_OP_GEMM || same_params(q[e]->ops[q[e]->pend_op]))) ++e;
    const int cnt = e - b;
    tln_program* p0 = q[b];
    switch (p0->pend_kind) {
      case TLN_OP_GEMM:
        rc = launch_pending_gemms(q + b, cnt, s);
        break;
      case TLN_OP_GN_PARTIALS: {
        tln_gn_partials_call c[kMaxGroup];
        for (int k = 0; k < cnt; ++k) c[k] = q[b + k]->pend_gn;
        rc = tln_groupnorm_partials_multi(c, cnt, p0->slots[o.s0.slot].cols, s);
        break;
      }
      case TLN_OP_POOL: {
        tln_pool_call c[kMaxGroup];
        for (int k = 0; k < cnt; ++k)
          c[k] = tln_pool_call{q[b + k]->lat, q[b + k]->d_dist, 4 * q[b + k]->N, q[b + k]->pend_pool_out};
        const float* w[4] = {o.p[0], o.p[1], o.p[2], o.p[3]};
        const float* bb[4] = {o.p[4], o.p[5], o.p[6], o.p[7]};
        int dims[6] = {o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], 0};
        tln_program* t0 = pp[0];   // stage timing of a group lives on its first program
        if (t0->timing) TLN_HIP(hipEventRecord(t0->tev[2], s));
        rc = tln_pointnet_pool_multi(c, cnt, p0->dist_cols, o.i[0], w, bb, dims, o.i[6], s);
        if (!rc && t0->timing) {
          TLN_HIP(hipEventRecord(t0->tev[3], s));
          t0->tset[1] = true;
        }
        break;
      }
      case TLN_OP_GRU: {
        tln_gru_call c[kMaxGroup];
        for (int k = 0; k < cnt; ++k) c[k] = q[b + k]->pend_gru;
        rc = tln_gru_cell_multi_opt(c, cnt, p0->slots[o.out].cols, o.p[0], o.p[1], o.p[2], o.p[3], &p0->opt, s);
        break;
      }
      case TLN_OP_AFLOW: {
        tln_aflow_call c[kMaxGroup];
        for (int k = 0; k < cnt; ++k) c[k] = q[b + k]->pend_aflow;
        rc = tln_aflow_multi(c, cnt, p0->slots[o.out].cols, o.f[0], o.f[1], o.f[2], o.i[0], o.bias, s);
        break;
      }
      case TLN_OP_SLICE_DEFORM: {
        tln_slice_call c[kMaxGroup];
        bool ls = p0->pend_slice.d_logsm != nullptr, mixed = false;
        for (int k = 0; k < cnt; ++k) {
          c[k] = q[b + k]->pend_slice;
          mixed = mixed || ((c[k].d_logsm != nullptr) != ls);
        }
        tln_program* t0 = pp[0];
        if (t0->timing && !t0->tset[2]) {
          TLN_HIP(hipEventRecord(t0->tev[4], s));
          t0->tset[2] = true;
        }
        const int cb = p0->slots[o.s0.slot].cols, ncls = p0->slots[o.s1.slot].cols;
        if (!mixed) {
          rc = tln_slice_deform_multi(c, cnt, cb, ncls, o.p[0], o.p[1], o.p[2], o.bias, s);
        } else {
          for (int k = 0; k < cnt && !rc; ++k) rc = tln_slice_deform_multi(c + k, 1, cb, ncls, o.p[0], o.p[1], o.p[2], o.bias, s);
        }
        if (!rc && t0->timing) TLN_HIP(hipEventRecord(t0->tev[5], s));
        break;
      }
      case TLN_OP_COPY: {
        // copies of one width in one launch; other widths (never, in one model) one by one
        int done = b;
        while (done < e) {
          CopyJobs jobs;
          const int cols = q[done]->pend_copy.cols;
          int64_t rmax = 0;
          int mm = 0;
          while (done + mm < e && q[done + mm]->pend_copy.cols == cols && mm < 8) {
            const auto& pc = q[done + mm]->pend_copy;
            jobs.j[mm].src = pc.src;
            jobs.j[mm].dst = pc.dst;
            jobs.j[mm].ld_src = pc.ld_src;
            jobs.j[mm].ld_dst = pc.ld_dst;
            jobs.j[mm].rows = pc.rows;
            jobs.j[mm].zero_row0 = pc.zero_row0;
            if (pc.rows > rmax) rmax = pc.rows;
            ++mm;
          }
          for (int i = mm; i < 8; ++i) jobs.j[i] = jobs.j[0];
          hipLaunchKernelGGL(k_copy_rows_multi, dim3((unsigned)tln_cdiv(rmax * (cols / 4), 256), (unsigned)mm), dim3(256), 0, s,
                             jobs, cols);
          TLN_LAUNCH_CHECK();
          done += mm;
        }
        break;
      }
      default:
        tln_set_error("group mode: op kind %d has no batched form", p0->pend_kind);
        rc = TLN_E_INVALID;
    }
    b = e;
  }
  for (int k = 0; k < n; ++k) {
    pp[k]->has_pending = false;
    pp[k]->pend_kind = 0;
  }
  return rc;
}

// ops [begin, end) of all programs: everything but the products per program, the products together
int walk_group(tln_program* const* pp, int n, int early, float* const* d_out, const int64_t* out_rows, int out_cols,
               hipStream_t s, int begin, int end, bool fresh) {
  int at[kMaxGroup];
  for (int k = 0; k < n; ++k) at[k] = begin;
  bool first = true;
  int rc = TLN_OK;
  for (;;) {
    bool any = false;
    for (int k = 0; k < n && !rc; ++k) {
      tln_program* p = pp[k];
      p->has_pending = false;
      if (at[k] >= end || (p->w_finished && !(fresh && first))) continue;
      p->defer = true;
      rc = walk(p, WALK_RUN, early, d_out[k], out_rows[k], out_cols, s, at[k], end, fresh && first);
      p->defer = false;
      at[k] = p->w_next;
      any = any || p->has_pending;
    }
    first = false;
    if (rc || !any) break;
    rc = launch_pending(pp, n, s);
    if (rc) break;
  }
  for (int k = 0; k < n; ++k) {
    pp[k]->defer = false;
    pp[k]->has_pending = false;
  }
  return rc;
}
}  // namespace

extern "C" int tln_program_run_group(tln_program_t* const* pp, int n, int early, float* const* d_out,
                                     const int64_t* out_rows, int out_cols, void* stream_) {
  TLN_REQUIRE(pp && d_out && out_rows && n >= 1 && n <= kMaxGroup, "bad group of %d programs", n);
  for (int k = 0; k < n; ++k) {
    TLN_REQUIRE(pp[k] && pp[k]->frame_open, "tln_program_run_group: program %d has no open frame", k);
    TLN_REQUIRE(pp[k]->ops.size() == pp[0]->ops.size() && pp[k]->split == pp[0]->split, "the programs differ");
    for (int j = 0; j < k; ++j) TLN_REQUIRE(pp[j] != pp[k], "the same program twice");
  }
  hipStream_t s = (hipStream_t)stream_;
  const int n_ops = (int)pp[0]->ops.size();
  int rc = TLN_OK;
  for (int k = 0; k < n && !rc; ++k) {
    rc = walk(pp[k], WALK_DRY, early, nullptr, out_rows[k], out_cols, s, 0, n_ops, true);
    if (!rc) rc = ensure_buf(pp[k]->arena, pp[k]->alloc.high + kAlign, s);
    pp[k]->calls.clear();
  }
  if (rc) return rc;
  rc = walk_group(pp, n, early, d_out, out_rows, out_cols, s, 0, pp[0]->split, true);
  {
    // the exact coarse counts of every sequence (one wait) and the coarse tables of all of them in one batch of launches
    tln_lattice_t* lats[kMaxGroup];
    for (int k = 0; k < n; ++k) lats[k] = pp[k]->lat;
    int rc2 = TLN_OK;
    if (group_batches(pp[0]->opt, 0)) rc2 = tln_lattice_prepare_levels_finish_multi(lats, n, s);
    else
      for (int k = 0; k < n && !rc2; ++k) rc2 = tln_lattice_prepare_levels_finish(lats[k], s);
    if (rc == TLN_OK) rc = rc2;
  }
  for (int k = 0; k < n; ++k) {
    tln_program* p = pp[k];
    if (rc == TLN_OK) {
      bool exceeded = false;
      rc = check_counts(p, &exceeded);
      if (rc == TLN_OK && exceeded && !p->w_finished)
        rc = replan(p, p->split, early, d_out[k], out_rows[k], out_cols, s);
    }
  }
  if (rc == TLN_OK) rc = walk_group(pp, n, early, d_out, out_rows, out_cols, s, pp[0]->split, n_ops, false);
  for (int k = 0; k < n; ++k) {
    if (rc == TLN_OK) commit_states(pp[k]);
    pp[k]->frame_open = false;
  }
  return rc;
}

extern "C" int tln_program_run_pair(tln_program_t* pa, tln_program_t* pb, int early, float* d_out_a, int64_t out_rows_a,
                                    float* d_out_b, int64_t out_rows_b, int out_cols, void* stream_) {
  tln_program_t* const pp[2] = {pa, pb};
  float* const outs[2] = {d_out_a, d_out_b};
  const int64_t rows[2] = {out_rows_a, out_rows_b};
  return tln_program_run_group(pp, 2, early, outs, rows, out_cols, stream_);
}

// the slice head (TLN_OP_SLICE_DEFORM) of the next frame also writes log_softmax(scores) to d_logsm [N, classes]
// (models.py:466-468 returns both); one-shot: cleared by the op that used it
extern "C" int tln_program_set_aux_out(tln_program_t* p, float* d_logsm) {
  TLN_REQUIRE(p, "null program");
  p->aux_out = d_logsm;
  return TLN_OK;
}

// ---- measurement: HIP events around K1 / K2 / K8 of a frame -----------------------------------------------------------
extern "C" int tln_program_timing(tln_program_t* p, int enable) {
  TLN_REQUIRE(p, "null program");
  if (enable)
    for (int i = 0; i < 6; ++i)
      if (!p->tev[i]) TLN_HIP(hipEventCreate(&p->tev[i]));
  p->timing = enable != 0;
  p->tset[0] = p->tset[1] = p->tset[2] = false;
  return TLN_OK;
}

// milliseconds of the last frame's K1 (distribute), K2 (pool), K8 (slice kernels); -1 where the frame had none
extern "C" int tln_program_timing_read(tln_program_t* p, float* ms_out) {
  TLN_REQUIRE(p && ms_out && p->timing, "stage timing is off");
  for (int k = 0; k < 3; ++k) {
    ms_out[k] = -1.f;
    if (!p->tset[k]) continue;
    TLN_HIP(hipEventSynchronize(p->tev[2 * k + 1]));
    TLN_HIP(hipEventElapsedTime(&ms_out[k], p->tev[2 * k], p->tev[2 * k + 1]));
  }
  return TLN_OK;
}

// ---- measurement: the gather-GEMM launches of the last frame, replayed back to back between two HIP events ------
extern "C" int tln_program_capture_gemms(tln_program_t* p, int enable) {
  TLN_REQUIRE(p, "null program");
  p->capture = enable != 0;
  if (!p->capture) p->calls.clear();
  return TLN_OK;
}

static int replay_one(const GemmCall& c, const tln_options& o, hipStream_t s) {
  const tln_gemm_call call{c.M, c.N, &c.a[0], c.two ? &c.a[1] : nullptr, c.w, c.w_is_nk, c.bias, c.res, c.ld_res,
                           c.relu, c.out, c.ld_out, c.stats};
  return tln_gather_gemm_opt(&call, &o, s);
}

// Device memory the program owns: [0] the arena of the frame's temporaries (capacity), [1] its high-water mark of the
// last frame, [2] the K1 output buffer (indices | weights | distributed rows when wanted), [3] the hidden-state buffers
// (two per fusion module), [4] total of 0, 2, 3.
extern "C" int tln_program_memory(const tln_program_t* p, int64_t* out) {
  TLN_REQUIRE(p && out, "null argument");
  out[0] = (int64_t)p->arena.bytes;
  out[1] = (int64_t)p->alloc.high;
  out[2] = (int64_t)p->k1.bytes;
  out[3] = 0;
  for (int st = 0; st < TLN_MAX_STATES; ++st)
    for (int k = 0; k < 2; ++k) out[3] += (int64_t)p->state_buf[st][k].bytes;
  out[4] = out[0] + out[2] + out[3];
  return TLN_OK;
}
// frames whose exact coarse vertex counts exceeded the planning prediction and were planned again (predict_bounds / replan)
extern "C" int64_t tln_program_replans(const tln_program_t* p) { return p ? p->replans : -1; }

extern "C" int tln_program_set_options(tln_program_t* p, const tln_options* opt) {
  TLN_REQUIRE(p, "null program");
  p->opt = tln_opt(opt);
  return TLN_OK;
}

extern "C" int tln_program_replay_gemms(tln_program_t* p, int reps, double* ms_total, int64_t* launches, double* flops,
                                        double* bytes, void* stream_) {
  TLN_REQUIRE(p && reps > 0 && ms_total && launches && flops && bytes, "bad replay arguments");
  hipStream_t s = (hipStream_t)stream_;
  *ms_total = 0.0;
  *launches = 0;
  *flops = 0.0;
  *bytes = 0.0;
  if (p->calls.empty()) return TLN_OK;
  hipEvent_t e0, e1;
  TLN_HIP(hipEventCreate(&e0));
  TLN_HIP(hipEventCreate(&e1));
  auto launch_all = [&]() -> int {
    for (const GemmCall& c : p->calls) {
      int rc = replay_one(c, p->opt, s);
      if (rc) return rc;
    }
    return TLN_OK;
  };
  int rc = launch_all();  // warm
  if (!rc) {
    TLN_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < reps && !rc; ++r) rc = launch_all();
    TLN_HIP(hipEventRecord(e1, s));
    TLN_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    TLN_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_total = ms;
  }
  if (!rc && getenv("TLN_GEMM_DUMP")) {  // per-launch table for tools/: 20 back-to-back launches of each call
    int idx = 0;
    for (const GemmCall& c : p->calls) {
      TLN_HIP(hipEventRecord(e0, s));
      for (int r = 0; r < 20; ++r) replay_one(c, p->opt, s);
      TLN_HIP(hipEventRecord(e1, s));
      TLN_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      TLN_HIP(hipEventElapsedTime(&ms, e0, e1));
      const double K = (double)c.a[0].taps * c.a[0].cin + (c.two ? (double)c.a[1].taps * c.a[1].cin : 0.0);
      fprintf(stderr, "gemm %2d M=%7ld N=%4d K=%5.0f taps=%d cin=%d two=%d gn=%d res=%d stats=%d  %.2f us  %.1f TF", idx++,
              (long)c.M, c.N, K, c.a[0].taps, c.a[0].cin, (int)c.two, c.a[0].d_gn_partials ? 1 : 0, c.res ? 1 : 0,
              c.stats ? 1 : 0, ms * 50.0, 2.0 * c.M * K * c.N / (ms * 50.0e-6) * 1e-12);
      if (atoi(getenv("TLN_GEMM_DUMP")) >= 2) {  // the direct kernel over its waves-per-tile choices
        fprintf(stderr, " | G:");
        for (int G = 1; G <= 12; ++G) {
          tln_options og = p->opt;
          og.gemm_groups = G;
          TLN_HIP(hipEventRecord(e0, s));
          for (int r = 0; r < 20; ++r) replay_one(c, og, s);
          TLN_HIP(hipEventRecord(e1, s));
          TLN_HIP(hipEventSynchronize(e1));
          TLN_HIP(hipEventElapsedTime(&ms, e0, e1));
          fprintf(stderr, " %d:%.1f", G, ms * 50.0);
        }
      }
      fprintf(stderr, "\n");
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  for (const GemmCall& c : p->calls) {
    const double K = (double)c.a[0].taps * c.a[0].cin + (c.two ? (double)c.a[1].taps * c.a[1].cin : 0.0);
    const double cin = (double)c.a[0].cin + (c.two ? (double)c.a[1].cin : 0.0);
    *flops += 2.0 * (double)c.M * K * c.N;
    // algorithmic bytes: every source row once, the output once, the weights, the tap table,
    // the residual if any
    *bytes += 4.0 * ((double)c.M * cin + (double)c.M * c.N + K * c.N) + 4.0 * (double)c.M * c.a[0].taps +
              (c.res ? 4.0 * (double)c.M * c.N : 0.0);
  }
  *launches = (int64_t)p->calls.size() * reps;
  *flops *= reps;
  *bytes *= reps;
  return TLN_OK;
}

// The products of the last frame of 1..8 lock-stepped programs replayed as the group issued them: product i of every
// program through ONE tln_gather_gemm_multi call (the launches of the timed mode of bench.py on one stream).
extern "C" int tln_program_replay_gemms_group(tln_program_t* const* pp, int n, int reps, double* ms_total, int64_t* launches,
                                              double* flops, double* bytes, void* stream_) {
  TLN_REQUIRE(pp && n >= 1 && n <= 8 && reps > 0 && ms_total && launches && flops && bytes, "bad replay arguments");
  hipStream_t s = (hipStream_t)stream_;
  *ms_total = 0.0;
  *launches = 0;
  *flops = 0.0;
  *bytes = 0.0;
  const size_t nc = pp[0]->calls.size();
  for (int k = 0; k < n; ++k) TLN_REQUIRE(pp[k] && pp[k]->calls.size() == nc, "the programs captured different products");
  if (nc == 0) return TLN_OK;
  hipEvent_t e0, e1;
  TLN_HIP(hipEventCreate(&e0));
  TLN_HIP(hipEventCreate(&e1));
  auto launch_all = [&]() -> int {
    for (size_t i = 0; i < nc; ++i) {
      tln_gemm_call calls[8];
      for (int k = 0; k < n; ++k) {
        const GemmCall& c = pp[k]->calls[i];
        calls[k] = tln_gemm_call{c.M, c.N, &c.a[0], c.two ? &c.a[1] : nullptr, c.w, c.w_is_nk, c.bias, c.res, c.ld_res,
                                 c.relu, c.out, c.ld_out, c.stats};
      }
      int rc = tln_gather_gemm_multi_opt(calls, n, &pp[0]->opt, s);
      if (rc) return rc;
    }
    return TLN_OK;
  };
  int rc = launch_all();  // warm
  if (!rc) {
    TLN_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < reps && !rc; ++r) rc = launch_all();
    TLN_HIP(hipEventRecord(e1, s));
    TLN_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    TLN_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_total = ms;
  }
  if (!rc && getenv("TLN_GEMM_DUMP")) {  // per-product table for tools/: 10 back-to-back shared launches of each product
    for (size_t i = 0; i < nc; ++i) {
      tln_gemm_call calls[8];
      double fl = 0.0;
      int64_t msum = 0, mmax = 0;
      for (int k = 0; k < n; ++k) {
        const GemmCall& c = pp[k]->calls[i];
        calls[k] = tln_gemm_call{c.M, c.N, &c.a[0], c.two ? &c.a[1] : nullptr, c.w, c.w_is_nk, c.bias, c.res, c.ld_res,
                                 c.relu, c.out, c.ld_out, c.stats};
        const double K = (double)c.a[0].taps * c.a[0].cin + (c.two ? (double)c.a[1].taps * c.a[1].cin : 0.0);
        fl += 2.0 * (double)c.M * K * c.N;
        msum += c.M;
        if (c.M > mmax) mmax = c.M;
      }
      TLN_HIP(hipEventRecord(e0, s));
      for (int r = 0; r < 10; ++r) tln_gather_gemm_multi_opt(calls, n, &pp[0]->opt, s);
      TLN_HIP(hipEventRecord(e1, s));
      TLN_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      TLN_HIP(hipEventElapsedTime(&ms, e0, e1));
      const GemmCall& c = pp[0]->calls[i];
      fprintf(stderr, "group gemm %3zu Msum=%7ld Mmax=%6ld N=%4d taps=%d cin=%4d two=%d gn=%d res=%d nk=%d  %8.2f us  %6.1f TF\n", i,
              (long)msum, (long)mmax, c.N, c.a[0].taps, c.a[0].cin, (int)c.two, c.a[0].d_gn_partials ? 1 : 0, c.res ? 1 : 0,
              c.w_is_nk, ms * 100.0, fl / (ms * 100.0e-6) * 1e-12);
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  for (int k = 0; k < n; ++k)
    for (const GemmCall& c : pp[k]->calls) {
      const double K = (double)c.a[0].taps * c.a[0].cin + (c.two ? (double)c.a[1].taps * c.a[1].cin : 0.0);
      const double cin = (double)c.a[0].cin + (c.two ? (double)c.a[1].cin : 0.0);
      *flops += 2.0 * (double)c.M * K * c.N;
      *bytes += 4.0 * ((double)c.M * cin + (double)c.M * c.N + K * c.N) + 4.0 * (double)c.M * c.a[0].taps +
                (c.res ? 4.0 * (double)c.M * c.N : 0.0);
    }
  *launches = (int64_t)nc * n * reps;   // products (a shared launch counts once per product it carries)
  *flops *= reps;
  *bytes *= reps;
  return TLN_OK;
}

// What the matrix cores execute for the captured products of one frame of n lock-stepped programs (n = 1: one sequence
// alone): every gather-GEMM kernel counts its 32 x 32 x 32 steps (gemm_v2 after skipping the K chunks of absent taps, all
// kernels including the rows and columns a tile pads) into a device counter during ONE extra pass of the launches.
extern "C" int tln_program_replay_executed(tln_program_t* const* pp, int n, double* flops_executed, void* stream_) {
  TLN_REQUIRE(pp && n >= 1 && n <= 8 && flops_executed, "bad replay arguments");
  hipStream_t s = (hipStream_t)stream_;
  *flops_executed = 0.0;
  const size_t nc = pp[0]->calls.size();
  for (int k = 0; k < n; ++k) TLN_REQUIRE(pp[k] && pp[k]->calls.size() == nc, "the programs captured different products");
  if (nc == 0) return TLN_OK;
  unsigned long long* d_cnt = nullptr;
  TLN_HIP(hipMalloc(&d_cnt, 16 * sizeof(unsigned long long)));
  int rc = TLN_OK;
  if (hipMemsetAsync(d_cnt, 0, 16 * sizeof(unsigned long long), s) != hipSuccess) rc = TLN_E_HIP;
  tln_options oc = pp[0]->opt;
  oc.gemm_stamps = d_cnt;
  for (size_t i = 0; i < nc && !rc; ++i) {
    tln_gemm_call calls[8];
    for (int k = 0; k < n; ++k) {
      const GemmCall& c = pp[k]->calls[i];
      calls[k] = tln_gemm_call{c.M, c.N, &c.a[0], c.two ? &c.a[1] : nullptr, c.w, c.w_is_nk, c.bias, c.res, c.ld_res,
                               c.relu, c.out, c.ld_out, c.stats};
    }
    rc = tln_gather_gemm_multi_opt(calls, n, &oc, s);
  }
  unsigned long long h[16] = {0};
  if (!rc && (hipStreamSynchronize(s) != hipSuccess ||
              hipMemcpy(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess))
    rc = TLN_E_HIP;
  (void)hipFree(d_cnt);
  if (rc) return rc;
  *flops_executed = (double)h[8] * 65536.0;   // 2 x 32 x 32 x 32
  return TLN_OK;
}

extern "C" int tln_program_frame_rows(tln_program_t* p, const float** d_distributed, const int32_t** d_indices,
                                      const float** d_weights, int64_t* rows, int* cols) {
  TLN_REQUIRE(p && p->d_idx, "no frame yet");
  if (d_distributed) *d_distributed = p->d_dist;   // NULL when the frame's pool read the vertex bins instead
  if (d_indices) *d_indices = p->idx_valid ? p->d_idx : nullptr;   // (NULL: the frame did not produce them)
  if (d_weights) *d_weights = p->d_w;
  if (rows) *rows = 4 * p->N;
  if (cols) *cols = p->dist_cols;
  return TLN_OK;
}

extern "C" int tln_program_state_info(const tln_program_t* p, int id, int64_t* rows, int* cols, int* exists) {
  TLN_REQUIRE(p && id >= 0 && id < p->n_states, "bad state id");
  if (rows) *rows = p->state_rows[id];
  if (cols) *cols = p->state_cols[id];
  if (exists) *exists = p->state_has[id] ? 1 : 0;
  return TLN_OK;
}

extern "C" int tln_program_state_get(tln_program_t* p, int id, float* d_out, void* stream_) {
  TLN_REQUIRE(p && id >= 0 && id < p->n_states && d_out, "bad state id");
  TLN_REQUIRE(p->state_has[id], "hidden state %d does not exist", id);
  const size_t bytes = (size_t)p->state_rows[id] * p->state_cols[id] * sizeof(float);
  if (bytes)
    TLN_HIP(hipMemcpyAsync(d_out, p->state_buf[id][p->state_cur[id]].p, bytes, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream_));
  return TLN_OK;
}

extern "C" int tln_program_state_set(tln_program_t* p, int id, const float* d_in, int64_t rows, void* stream_) {
  TLN_REQUIRE(p && id >= 0 && id < p->n_states && d_in && rows >= 0, "bad state id");
  int cols = p->state_cols[id];
  if (cols == 0)
    for (const tln_slot& s : p->slots)
      if (s.kind == TLN_SLOT_STATE_NEW && s.state == id) cols = s.cols;
  TLN_REQUIRE(cols > 0, "hidden state %d has no width", id);
  hipStream_t s = (hipStream_t)stream_;
  TLN_REQUIRE(!p->seg_open || (p->state_has[id] && p->state_rows[id] == rows),
              "hidden state %d arrives with %lld rows inside a segmented run that was sized for %lld", id, (long long)rows,
              (long long)p->state_rows[id]);
  const size_t bytes = (size_t)rows * cols * sizeof(float);
  DevBuf& b = p->state_buf[id][p->state_cur[id]];
  int rc = ensure_buf(b, bytes ? bytes : kAlign, s);
  if (rc) return rc;
  if (bytes) TLN_HIP(hipMemcpyAsync(b.p, d_in, bytes, hipMemcpyDeviceToDevice, s));
  p->state_rows[id] = rows;
  p->state_cols[id] = cols;
  p->state_has[id] = true;
  return TLN_OK;
}
