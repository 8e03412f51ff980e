// K3+K4 for LARGE M (a level-0 lattice of a SemanticKITTI-sized scan: 2-3 x 10^4 vertices and more).
//
// Same product as gemm.hip (reference call sites lattice_modules.py:301/440/573, models.py:353/398, the 1x1 linears
// and the GRU projections lm:47-62): out[M,N] = prologue(gather(src, table))[M, taps*cin] @ W (+bias, +residual, ReLU),
// `v_mfma_f32_32x32x2_f32` (exact fp32).  What changes is who fetches what.  At M ~ 3 x 10^4 the direct kernel
// (one wave per 32x32 tile) gathers every source row N/32 times and the 64x64-tile kernel N/64 times, each wave
// through its own vector-memory path: the gather, not the matrix pipe, sets their pace (80 / 62 TFLOP/s on
// 192 -> 192).  Here a block owns 128 rows x ALL its columns (up to 192 per block):
//   * A (gathered rows) and B (weights) are staged ONCE per block and K chunk by LDS-DMA (`buffer_load_dwordx4 ... lds`,
//     16 B per lane, no VGPR round trip; a piece's address = buffer descriptor + one 32-bit lane offset + a scalar
//     offset, a missing neighbour = an offset out of the buffer's range, whose check writes the zero row): A as 128-byte
//     row pieces [BM][32 floats] whose 16-byte slots are XOR-swizzled with the row ((row>>1)&7, applied on the SOURCE
//     offset since the DMA image is lane-linear), so that the operand reads are conflict-free `ds_read_b128`; B as
//     [32][BN] ([K,N] weights, `ds_read_b32`) or as [BN][32] ([N,K] weights, like A);
//   * a ring of two LDS stages (three by TLN_V2_STAGES2), ONE barrier per chunk, behind the first MFMA group of the
//     chunk's last step: `s_waitcnt vmcnt` (this thread's DMAs of the next chunk landed) -> `s_barrier` -> the DMAs of
//     chunk t + STAGES into this chunk's stage, the first fragments of chunk t + 1 under the remaining MFMAs;
//   * eight waves as 4 x 2 (four as 4 x 1 for narrow N, 2 x 2 on 64-row tiles), each 32 x 96 / 32 x 64 / 32 x 32
//     outputs; the k order inside a chunk is permuted (lane half h takes k = 8j + 4h + e), identically for A and B, so
//     one b128 read feeds 4 MFMAs;
//   * the K loop is built for `64 cycles per MFMA + ~3 per vector instruction` (tools/micro/mfma_rate.hip: nothing hides
//     in an fp32 MFMA's shadow): scheduled by hand in groups behind scheduling fences, every LDS address a base register
//     + immediate (chunk body instantiated per ring stage), 44 vector instructions per 48 MFMAs;
//   * the GroupNorm affine + ReLU of the consumer side (GN -> ReLU -> conv) is applied to the NEXT step's fragments behind
//     the current step's last MFMA group, a missing neighbour stays an exact zero row: fma, then ONE v_med3_f32 that
//     clamps to [0, inf) or to [0, 0];
//   * epilogue: bias, residual, ReLU by buffer loads / stores from the rows' byte offsets, per-32-row (sum, sum^2) in
//     fp64 for the next GroupNorm.
// One source only (the two-source products live on small levels); rows past the source read as zeros (pad = 0).
#include "gemm_args.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x8 __attribute__((ext_vector_type(8)));

#define V2_STAGES 3
#ifndef V2_FOLD
#define V2_FOLD 7   // split accumulation (see v2_body) per column-tile count of a wave: bit 0 TN = 1, bit 1 TN = 2, bit 2 TN = 3;
#endif              // 0: one chain over all of K (measurement builds: TLN_EXTRA_FLAGS=-DV2_FOLD=3 ...)

// GRU = true (only <4,2,1,3, W_NK, !PRO>, N = 3C, C a multiple of 64): the whole GRU cell (K9) as ONE product with two
// sources: the K loop runs over the C channels of x (weights W_ih, g.W) and then over the C channels of h (W_hh, g.W2;
// rows past s[1].src_rows are the zero padding of lm:59-60).  The 192 columns of a block are 64 channels x 3 gates,
// ordered so that the three 32-column tiles of a WAVE are the r, z, n gates of the same 32 channels (column t of the
// block: wave t/96, gate (t%96)/32, channel 64*blockIdx.y + 32*(t/96) + t%32; the weight row staged for it is
// gate*C + channel): r, z, n of one (row, channel) sit at the same accumulator index of the wave's tiles.  r and z need
// only gi + gh — one accumulator over both sources —, n needs gi_n and gh_n apart: a fourth accumulator tile takes the
// h chunks of the n gate.  The cell is the epilogue: neither gi nor gh reaches memory, no gates kernel.
// g.bias = b_ih, g.bias2 = b_hh, g.out = h' [M, C].
// STAGES = 3: DMAs two chunks ahead; STAGES = 2 (128 x 128 tile): one chunk ahead, but two workgroups fit a CU's LDS
template <int WM, int WN, int TM, int TN, bool W_NK, bool PRO, bool GRU = false, int STAGES = V2_STAGES>
__device__ __forceinline__ void v2_body(const GemmArgs& g, const int bx, const int by) {
#if __HIP_DEVICE_COMPILE__   // (the buffer-resource type of the LDS-DMAs exists in the device pass only)
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  // (B: a thread count that does not divide the 16-byte pieces of the weight tile — the 96-row tile's 384 threads on a
  // 128-column tile — rounds the pieces per thread up; the surplus pieces read out of the buffer's range: zeros into slack behind the tile)
  constexpr int A_PIECES = BM * 8 / NT, B_PIECES = (BN * 8 + NT - 1) / NT;
  constexpr bool B_PAD = (BN * 8) % NT != 0;
  constexpr int A_BYTES = BM * 128, B_BYTES = B_PIECES * NT * 16;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int PIECES = A_PIECES + B_PIECES;
  static_assert((BM * 8) % NT == 0, "whole A pieces per thread");
  static_assert(PIECES < 32, "vmcnt immediate");

  extern __shared__ __attribute__((aligned(16))) char smem2[];
  char* ring = smem2;
  int* Is = reinterpret_cast<int*>(smem2 + STAGES * STAGE);            // [BM][taps]
  const SrcDev& s = g.s[0];
  const int taps = s.taps;
  float* GS = reinterpret_cast<float*>(Is + BM * TLN_TAPS);   // [cin / 4][8]: GroupNorm scale | shift of four channels
  int* Rid = reinterpret_cast<int*>(GS + 2 * s.cin);   // [BM] row of block position r (the product's row order, or m0 + r)
  int* Taps = Rid + BM;                             // [0] = number of taps present in this block, [1..] = which

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int l31 = lane & 31, half = lane >> 5;
  // (blocks start in index order: with a launch order the heaviest blocks — most taps present — come first)
  // (64-row tiles: the two halves of the 128-row tiles the launch order is made of, in that order)
  const int64_t m0 = (BM == 128 && s.order) ? (int64_t)s.order[bx] * BM
                     : (BM == 64 && s.order) ? (int64_t)s.order[bx >> 1] * 128 + (bx & 1) * 64
                                             : (int64_t)bx * BM;
  if (m0 >= g.M) return;   // (uniform: the second half of a last 128-row tile, a grid sized for a longer product)
  const int n0 = by * BN;
  const bool has_table = s.table != nullptr;
  const int src_rows = (int)s.src_rows;
  const int ld_bytes = (int)s.ld * 4;

  // ---- block prologue: the rows of this block (in the table's row order when there is one: rows with the same set of
  // present taps sit together, lattice.hip), their tap indices (or the row's own index), the GroupNorm scale / shift,
  // and which taps at least one row of the block has (a K chunk of a tap nobody has multiplies zeros: skipped)
  const int32_t* __restrict__ perm = s.perm;
  for (int r = tid; r < BM; r += NT) {
    const int64_t p = m0 + r;
    // (kept as the row's BYTE offset in the output — host: M * ld_out * 4 < 2^31, a residual has the output's row
    //  stride —, a position past the end as INT_MIN: the epilogue's loads and stores are buffer operations whose range
    //  check drops them, one add per element instead of a 64-bit multiply-add)
    Rid[r] = p < g.M ? (perm ? perm[p] : (int)p) * ((int)g.ld_out * 4) : (int)0x80000000;
  }
  unsigned present = 0;                       // taps seen by this wave
  for (int i0 = 0; i0 < BM * taps; i0 += NT) {
    const int i = i0 + tid;
    const int r = i / taps, tap = i - r * taps;
    int idx = -1;
    if (i < BM * taps) {
      const int64_t p = m0 + r;
      const int m = p < g.M ? (perm ? perm[p] : (int)p) : -1;
      if (m >= 0) idx = has_table ? s.table[(int64_t)m * taps + tap] : m;
      if (idx >= src_rows) idx = -1;          // rows past the source: zeros (pad value 0, checked on the host)
      if (GRU && tap == 1 && idx >= (int)g.s[1].src_rows) idx = -1;   // "tap" 1 = the row of h
      // kept as the row's BYTE offset in the source (host: rows * ld * 4 < 2^31), a missing row as INT_MIN: far outside
      // the buffer's range, whose check then delivers the zero row (tools/micro/buffer_lds_oob.hip) — no select, no
      // 64-bit address arithmetic in the K loop
      Is[i] = idx < 0 ? (int)0x80000000 : idx * ld_bytes;
    }
    if (has_table && !GRU) {
#pragma unroll
      for (int k = 0; k < TLN_TAPS; ++k)
        if (__ballot(idx >= 0 && tap == k) != 0ull) present |= 1u << k;
    }
  }
  if (lane == 0) Taps[16 + wv] = (int)present;   // [16 .. 16 + waves): what each wave saw
  if (PRO) {   // scale and shift of four channels side by side: [c / 4][scale x 4 | shift x 4]
    for (int c = tid; c < s.cin; c += NT) {
      GS[(c >> 2) * 8 + (c & 3)] = s.scale[c];
      GS[(c >> 2) * 8 + 4 + (c & 3)] = s.shift[c];
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  // (decided in uniform code: read under the divergent branch below, the arguments of a shared launch came by vector loads
  //  and cost the GRU kernel 14 registers.  fold_taps: the GRU launch's switch for the skip, host)
  const bool gru_skip_h = GRU && g.fold_taps != 0 && m0 >= (int64_t)g.s[1].src_rows;
  // every wave builds the same list of present taps (identical values to the same words: no further barrier)
  if (lane == 0) {
    unsigned mask = 0;
    for (int w = 0; w < WM * WN; ++w) mask |= (unsigned)Taps[16 + w];
    if (!has_table || GRU) mask = (1u << taps) - 1u;   // only tap tables have holes worth skipping
    // GRU: a block whose rows all lie past the hidden state (vertices born in this frame: 8-22 % of a frame's rows on
    // the headline workload, at the end of the vertex order) has nothing but zero rows of h: its h chunks — half its K
    // loop — would add exact zeros and are skipped (same bits)
    if (gru_skip_h) mask = 1u;
    int n_present = 0, prev_group = -1;
    for (int k = 0; k < taps; ++k)
      if ((mask >> k) & 1u) {
        // [23 + i], i = 1..8 (words 24..31: behind the waves' masks): the accumulators are folded before the i-th present
        // tap (split accumulation below): its group of g.fold_taps LOGICAL taps differs from the previous present tap's
        const int grp = k / (g.fold_taps > 0 ? g.fold_taps : 1);
        if (n_present > 0) Taps[23 + n_present] = grp != prev_group ? 1 : 0;
        prev_group = grp;
        Taps[1 + n_present++] = k;
      }
    Taps[0] = n_present;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int cpt = s.cin >> 5;                 // chunks per tap
  const int nchunks = Taps[0] * cpt;
  // measurement (tln_program_replay_executed): the 32 x 32 x 32 steps this block's matrix cores execute
  if (!GRU && g.dbg && tid == 0) atomicAdd(&g.dbg[8], (unsigned long long)nchunks * (TM * WM) * (TN * WN));

  // The LDS-DMAs of one chunk: every thread PIECES x 16 bytes, lane-linear pieces of 1 KiB per wave, as
  // `buffer_load_dwordx4 ... lds`: the address of a piece is a buffer (source rows / weights) + ONE 32-bit lane offset +
  // a scalar offset.  What depends on the lane alone is computed here, once per block; per chunk a thread adds its rows'
  // byte offsets (from the index list) to its pieces' constants — three vector instructions where the flat-address
  // version had fifty-five (a vector instruction costs the matrix pipe ~3 cycles: tools/micro/mfma_rate.hip).
  const unsigned RSRC3 = 0x00020000u;        // raw buffer, 32-bit data format
  const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)s.src, 0, (int)(s.src_rows * s.ld * 4), RSRC3);
  const __amdgpu_buffer_rsrc_t rA1 =
      GRU ? __builtin_amdgcn_make_buffer_rsrc((void*)g.s[1].src, 0, (int)(g.s[1].src_rows * s.ld * 4), RSRC3) : rA0;
  const int w_bytes = (int)((W_NK ? (int64_t)g.N : (int64_t)(GRU ? s.cin : g.K0)) * g.ldw * 4);
  const __amdgpu_buffer_rsrc_t rB0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, w_bytes, RSRC3);
  const __amdgpu_buffer_rsrc_t rB1 = GRU ? __builtin_amdgcn_make_buffer_rsrc((void*)g.W2, 0, w_bytes, RSRC3) : rB0;
  unsigned a_lane[A_PIECES], b_lane[B_PIECES];
#pragma unroll
  for (int p = 0; p < A_PIECES; ++p) {
    const int e = p * NT + tid;
    const int r = e >> 3, q = e & 7;
    a_lane[p] = 16u * (unsigned)(q ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int p = 0; p < B_PIECES; ++p) {
    const int e = p * NT + tid;
    bool ok;
    unsigned off;
    if (!W_NK) {
      const int k = e / (BN / 4), nq = e - k * (BN / 4);
      const int n = n0 + 4 * nq;
      ok = n < g.N && (!B_PAD || e < BN * 8);
      off = 4u * ((unsigned)(B_PAD ? (k & 31) : k) * (unsigned)g.ldw + (unsigned)n);
    } else {
      const int r = e >> 3, q = e & 7;
      int n = n0 + r;
      if (GRU) n = ((r % 96) >> 5) * s.cin + 32 * WN * by + 32 * (r / 96) + (r & 31);
      ok = n < g.N && (!B_PAD || e < BN * 8);
      off = 4u * ((unsigned)n * (unsigned)g.ldw + 4u * (unsigned)(q ^ ((r >> 1) & 7)));
    }
    b_lane[p] = ok ? off : 0x80000000u;      // (columns past N: zeros from the range check)
  }
  const unsigned is_lane = 4u * (unsigned)((tid >> 3) * taps);   // this thread's row of piece 0 in the index list; piece p: + 64 p rows
  struct Dma {
    int tap, c0;
    unsigned voff[A_PIECES];
  };
  // past the last chunk the DMAs repeat the last one into a stage nobody reads any more: no branch in the loop body
  auto dma_tap = [&](int t_raw, Dma& d) {
    const int t = t_raw < nchunks ? t_raw : nchunks - 1;
    const int ti = t / cpt;
    d.tap = Taps[1 + ti];                    // the ti-th tap present in this block (uniform; made a scalar in dma_rows)
    d.c0 = (t - ti * cpt) << 5;
  };
  auto dma_rows = [&](Dma& d) {
    d.tap = __builtin_amdgcn_readfirstlane(d.tap);   // (the DMAs want it in scalar registers: buffer choice, scalar offset)
    const char* isp = reinterpret_cast<const char*>(Is) + is_lane + 4 * d.tap;
#pragma unroll
    for (int p = 0; p < A_PIECES; ++p)
      d.voff[p] = (unsigned)*reinterpret_cast<const int*>(isp + p * (NT / 8) * taps * 4) + a_lane[p];
  };
  auto dma_piece = [&](const Dma& d, int st, int piece) {   // st, piece: compile-time after unrolling
    char* As = ring + st * STAGE;
    char* Bs = As + A_BYTES;
    if (piece < A_PIECES) {
      const int p = piece;
      __builtin_amdgcn_raw_ptr_buffer_load_lds((GRU && d.tap) ? rA1 : rA0,
                                               (__attribute__((address_space(3))) void*)(As + (p * NT + wv * 64) * 16), 16,
                                               d.voff[p], 4 * d.c0, 0, 0);
    } else {
      const int p = piece - A_PIECES;
      // (GRU: "tap" 1 = the second source h with its own weights, both [3C][C])
      const int kbase = GRU ? d.c0 : d.tap * s.cin + d.c0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds((GRU && d.tap) ? rB1 : rB0,
                                               (__attribute__((address_space(3))) void*)(Bs + (p * NT + wv * 64) * 16), 16,
                                               b_lane[p], W_NK ? 4 * kbase : 4 * kbase * (int)g.ldw, 0, 0);
    }
  };
  auto issue = [&](int t_raw, int st) {       // a whole chunk at once (the ring's first fill)
    Dma d;
    dma_tap(t_raw, d);
    dma_rows(d);
#pragma unroll
    for (int piece = 0; piece < PIECES; ++piece) dma_piece(d, st, piece);
  };

  constexpr int TNA = GRU ? TN + 1 : TN;   // GRU: tile TN takes the h chunks of the n gate (gh_n apart from gi_n)
  f32x16 acc[TM][TNA];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNA; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Split accumulation (round 4).  One fp32 accumulation chain over all of K = taps * cin (1728 steps for 192 -> 192 x 9)
  // carries ~sqrt(K / K_block) times the rounding noise of a blocked sum: measured against a float64 evaluation of the
  // whole model (tools/parity64.py) the logits were 1.55 x as far from the truth as the fp32 CPU
  // restatement's (whose GEMMs are K-blocked), 1.08 x with the K range in three partial sums, 0.82 x in nine.  So the
  // accumulators are folded into `tot` whenever the chunk sequence crosses into another GROUP OF TAPS (V2_TAPS_PER_FOLD
  // logical taps, ~192-256 k values) and restart from zero.  The groups are defined on the logical tap ids, not on the
  // executed chunk count: a tap this block skips contributes an exact zero to its group's partial sum, so the result
  // stays independent of the row order / tap skipping, as before.
  // Only the [K,N]-weight instantiations fold: those are the 9-tap products (convolutions, coarsen, finefy); the [N,K]
  // ones are 1 x 1 linears and GRU projections (K <= 2 x 256: a short chain) and keep their registers.
  constexpr bool FOLD = !GRU && !W_NK && ((V2_FOLD >> (TN - 1)) & 1);
  f32x16 tot[FOLD ? TM : 1][FOLD ? TN : 1];
  if constexpr (FOLD) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.0f;
  }

  const int arow0 = wm * 32 * TM + l31;       // this lane's A row of tile 0 inside the block
  const int bcol0 = wn * 32 * TN + l31;       // this lane's B column of tile 0 inside the block

  // One chunk = four steps of 8 k (lane half h holds k = 8j + 4h + e, e = 0..3), one step = four groups of TM*TN MFMAs
  // (one per e).  The K loop is scheduled BY HAND, group by group, with a scheduling fence behind every group: left to
  // the compiler (even under sched_group_barrier hints) the body came out as runs of 15 bare MFMAs, a lump of 69 other
  // instructions around the DMAs and nine full `s_waitcnt lgkmcnt(0)` drains per chunk — the matrix pipe was busy 75 % of
  // the loop (tools/v2_stamps.py).  The plan per group: its MFMAs first, then — in the shadow of those MFMAs — a quarter
  // of the NEXT step's fragment loads; the GroupNorm transform of the next step's A values behind the last group of a
  // step, when their loads are half a step old.  The chunk barrier sits behind the FIRST group of the last step: by then
  // every read of this chunk's stage has been issued (the last step's fragments were loaded during the step before), so
  // behind it the DMAs of chunk t + STAGES may overwrite the stage and the first fragments of chunk t + 1 are loaded
  // under the remaining three groups — no chunk starts with an exposed LDS round trip.
  struct Frag {
    f32x4 a[TM];
    float b[TN][4];
    f32x4 sc, sh;   // PRO: the GroupNorm scale / shift of the step's four channels
  };
  // Where this lane's fragments sit, per ring stage: computed once, kept in registers (made opaque, or the compiler would
  // rather re-derive them with an add in front of every read — 45 of the loop's 120 vector instructions).  With the stage
  // a compile-time constant of the chunk body (the loop dispatches on it) every LDS read of the loop is base register +
  // immediate.  A: [row][32 floats], 16-byte slots swizzled with the row; B: [32][BN] or, for [N,K] weights, like A.
  unsigned a_at[STAGES][TM][4], b_at[STAGES][W_NK ? 4 : 1];
#pragma unroll
  for (int st = 0; st < STAGES; ++st) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = arow0 + 32 * i;
        a_at[st][i][j] = (unsigned)(size_t)(ring + st * STAGE + r * 128 + (((2 * j + half) ^ ((r >> 1) & 7)) << 4));
        asm volatile("" : "+v"(a_at[st][i][j]));
      }
    if (!W_NK) {
      b_at[st][0] = (unsigned)(size_t)(ring + st * STAGE + A_BYTES + (4 * half * BN + bcol0) * 4);
      asm volatile("" : "+v"(b_at[st][0]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // (the swizzle of a column's slot does not depend on the column tile: 32 jn rows further)
        b_at[st][j] = (unsigned)(size_t)(ring + st * STAGE + A_BYTES + bcol0 * 128 + (((2 * j + half) ^ ((bcol0 >> 1) & 7)) << 4));
        asm volatile("" : "+v"(b_at[st][j]));
      }
    }
  }
  const unsigned gs_lane = (unsigned)(size_t)GS + 32u * (unsigned)half;   // channel c0 + 8 j + 4 half: + 8 c0 + 64 j bytes
  auto lds_f4 = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)addr); };
  // (volatile: two single reads merged into one ds_read2 reach only 1 KB past their base and cost an add — a vector
  //  instruction, ~3 cycles of matrix pipe — where a ds_read_b32 with its 16-bit offset costs nothing)
  auto lds_f1 = [](unsigned addr) { return *reinterpret_cast<const volatile __attribute__((address_space(3))) float*>((size_t)addr); };
  // a quarter of a step's fragment loads: part 0 = A (+ scale / shift), parts 1..3 = the B values.  st, j, part: constants
  auto frag_load_part = [&](int st, int j, unsigned gs_c0, Frag& f, int part) {
    if (part == 0) {
      if (PRO) {
        f.sc = lds_f4(gs_c0 + 64 * j);
        f.sh = lds_f4(gs_c0 + 64 * j + 16);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a[i] = lds_f4(a_at[st][i][j]);
      return;
    }
    if (!W_NK) {
#pragma unroll
      for (int q = 0; q < 4 * TN; ++q) {        // 4 TN single values over the parts 1..3, in the order the MFMAs want them
        if (1 + (q * 3) / (4 * TN) != part) continue;
        const int e = q / TN, jn = q - e * TN;
        f.b[jn][e] = lds_f1(b_at[st][0] + ((8 * j + e) * BN + 32 * jn) * 4);
      }
    } else {
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        if (1 + (jn * 2) / TN != part) continue;   // (parts 1 and 2: the step's first group needs every column tile)
        const f32x4 v = lds_f4(b_at[st][j] + 32 * jn * 128);
        f.b[jn][0] = v[0];
        f.b[jn][1] = v[1];
        f.b[jn][2] = v[2];
        f.b[jn][3] = v[3];
      }
    }
  };
  // GroupNorm affine; ReLU and the zero row of a missing neighbour in ONE median: a clamp to [0, inf) for a row that
  // exists, to [0, 0] for one that does not
  auto frag_pro = [&](const float (&hi)[TM], Frag& f) {
    if (PRO) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) f.a[i][e] = __builtin_amdgcn_fmed3f(fmaf(f.a[i][e], f.sc[e], f.sh[e]), 0.0f, hi[i]);
    }
  };
  auto frag_mma_e = [&](const Frag& f, int e) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[jn][e], acc[i][jn], 0, 0, 0);
  };
  // PRO: upper end of the clamp behind the affine for this lane's rows in chunk t: inf for a row that exists under the
  // chunk's tap, 0 for a missing neighbour.  Two dependent LDS reads (tap list, index list)
  auto clamp_of = [&](int t_raw, float (&hi)[TM]) {
    const int t = t_raw < nchunks ? t_raw : nchunks - 1;
    const int tap = Taps[1 + t / cpt];
#pragma unroll
    for (int i = 0; i < TM; ++i) hi[i] = (PRO && Is[(arow0 + 32 * i) * taps + tap] < 0) ? 0.0f : __builtin_inff();
  };
#define V2_FENCE() __builtin_amdgcn_sched_barrier(0)

  // ---- the ring's first fill; the first chunk's first fragments
  Frag f0, f1;
  float hi[TM];
  if (nchunks > 0) {
#pragma unroll
    for (int k = 0; k < STAGES; ++k) issue(k, k);
    clamp_of(0, hi);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((STAGES - 1) * PIECES) : "memory");
#pragma unroll
    for (int part = 0; part < 4; ++part) frag_load_part(0, 0, gs_lane, f0, part);
    frag_pro(hi, f0);
  }
#ifdef TLN_V2_STAMPS
  unsigned long long st_dma = 0, st_bar = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long st_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
  // split accumulation: `tot += acc; acc = 0` in front of the first chunk of a present tap that opens another group of
  // logical taps (flags in Taps[24 ..], block prologue).  Checked BETWEEN chunk bodies with two scalar counters — inside
  // the chunk body the check made the compiler emit five bodies instead of three and a division per chunk.
  int fold_tap = 0, fold_next = cpt;          // the tap the next chunk belongs to, the chunk index where the next tap starts
  auto fold_check = [&](int t) {
    if constexpr (FOLD) {
      if (t == fold_next) {
        ++fold_tap;
        fold_next += cpt;
        if (__builtin_amdgcn_readfirstlane(Taps[23 + fold_tap])) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              tot[i][j] += acc[i][j];
              // (zeroed as eight 64-bit moves: left to the compiler it is sixteen 32-bit ones, and beside an fp32 MFMA
              //  every vector instruction costs the matrix pipe ~3 cycles)
              f64x8 z;
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                double d;
                asm volatile("v_mov_b64_e32 %0, 0" : "=v"(d));
                z[q] = d;
              }
              acc[i][j] = __builtin_bit_cast(f32x16, z);
            }
        }
      }
    }
  };
  // the body of chunk t in ring stage ST (a compile-time constant: every LDS address below is register + immediate)
  auto chunk = [&](int t, auto stage) {
    // GRU: the chunks of x come first, then those of h.  The n gate needs gi_n and gh_n apart: at the first chunk of h the
    // n-gate tile (x's share, complete) moves to the extra tile and starts again from zero for h's share — once per block,
    // a uniform branch; the K loop itself is the same for both sources
    if (GRU && t == cpt) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][TNA - 1][r] = acc[i][TN - 1][r];
          acc[i][TN - 1][r] = 0.0f;
        }
    }
    constexpr int st = decltype(stage)::value;
    constexpr int stn = st == STAGES - 1 ? 0 : st + 1;
    const int ti = t / cpt;
    const unsigned gs_c0 = gs_lane + 8u * (unsigned)((t - ti * cpt) << 5);
    const int tn = t + 1 < nchunks ? t + 1 : nchunks - 1;
    const unsigned gs_c0n = gs_lane + 8u * (unsigned)((tn - (tn / cpt) * cpt) << 5);
    Dma d;
    float hin[TM];
    int tapn = 0, isn[TM];
    // steps 0..2: the groups of step j, the loads of step j + 1 behind them
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Frag& cur = (j & 1) ? f1 : f0;
      Frag& nxt = (j & 1) ? f0 : f1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        frag_mma_e(cur, e);
        if (e == 3) frag_pro(hi, nxt);                         // (its operands came with part 0, three groups ago)
        frag_load_part(st, j + 1, gs_c0, nxt, e);
        if (j == 0 && e == 1) dma_tap(t + STAGES, d);         // (the DMAs' two dependent LDS reads, a step apart)
        if (j == 1 && e == 1) dma_rows(d);
        if (j == 0 && e == 2 && PRO) tapn = Taps[1 + tn / cpt];   // (the next chunk's clamp: two dependent reads as well)
        if (j == 1 && e == 2 && PRO) {
#pragma unroll
          for (int i = 0; i < TM; ++i) isn[i] = Is[(arow0 + 32 * i) * taps + tapn];
        }
        if (j == 2 && e == 2 && PRO) {
#pragma unroll
          for (int i = 0; i < TM; ++i) hin[i] = isn[i] < 0 ? 0.0f : __builtin_inff();
        }
        V2_FENCE();
      }
    }
    // step 3: behind its first group the chunk barrier — this thread's DMAs of chunk t + 1 have landed (those of the
    // chunks behind it may still fly), after the barrier everybody's have, and every read of this chunk's stage is done
    frag_mma_e(f1, 0);
#ifdef TLN_V2_STAMPS
    const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((STAGES - 2) * PIECES) : "memory");
    const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_barrier" ::: "memory");
    const unsigned long long tb2 = __builtin_amdgcn_s_memtime();
    st_dma += tb1 - tb0;
    st_bar += tb2 - tb1;
    if (g.dbg && lane == 0 && bx == 37 && by == 0 && blockIdx.z == 0 && t < 60) {   // one block's timeline
      g.dbg[32 + (t * 8 + wv) * 2] = tb1;
      g.dbg[32 + (t * 8 + wv) * 2 + 1] = tb2;
    }
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((STAGES - 2) * PIECES) : "memory");
#endif
    V2_FENCE();
#pragma unroll
    for (int e = 1; e < 4; ++e) {
      frag_mma_e(f1, e);
      // the DMAs of chunk t + STAGES into this chunk's stage, a third behind each group; the first fragments of chunk t + 1
#pragma unroll
      for (int piece = 0; piece < PIECES; ++piece)
        if ((piece * 3) / PIECES == e - 1) dma_piece(d, st, piece);
      if (e == 3) frag_pro(PRO ? hin : hi, f0);
      if (e == 1) frag_load_part(stn, 0, gs_c0n, f0, 0);
      frag_load_part(stn, 0, gs_c0n, f0, e);
      V2_FENCE();
    }
    if (PRO) {
#pragma unroll
      for (int i = 0; i < TM; ++i) hi[i] = hin[i];
    }
  };
  // STAGES chunks per trip, straight-line (a dispatch per chunk would meet in one loop head and pay ~50 register moves
  // per chunk for it), then the one or two that are left
  {
    int t = 0;
    const bool cpt_odd = (cpt & 1) != 0;     // (chunks per tap 2, 4, 6, 8 on this workload: a tap starts at an even chunk)
    for (; t + STAGES <= nchunks; t += STAGES) {
      fold_check(t);
      chunk(t, std::integral_constant<int, 0>{});
      if (STAGES == 3 || cpt_odd) fold_check(t + 1);
      chunk(t + 1, std::integral_constant<int, 1>{});
      if constexpr (STAGES == 3) {
        fold_check(t + 2);
        chunk(t + 2, std::integral_constant<int, 2>{});
      }
    }
    if (t < nchunks) {
      fold_check(t);
      chunk(t, std::integral_constant<int, 0>{});
    }
    if (STAGES == 3 && t + 1 < nchunks) {
      fold_check(t + 1);
      chunk(t + 1, std::integral_constant<int, 1>{});
    }
  }
#ifdef TLN_V2_STAMPS
  if (g.dbg && lane == 0 && by == 0 && blockIdx.z == 0 && (bx % 37) == 0) {   // a sample of blocks, every wave
    const unsigned long long st_end = __builtin_amdgcn_s_memtime();
    atomicAdd(&g.dbg[16], st_end - st_begin);
    atomicAdd(&g.dbg[17], st_dma);
    atomicAdd(&g.dbg[18], st_bar);
    atomicAdd(&g.dbg[19], (unsigned long long)nchunks);
    atomicAdd(&g.dbg[20], 1ull);
    atomicAdd(&g.dbg[21], __builtin_amdgcn_s_memrealtime() - st_rbegin);   // 100 MHz
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated DMAs of the last two rounds
  if constexpr (GRU) {
    if (nchunks == cpt) {   // (uniform) the h chunks were skipped: x's share of the n gate still sits in tile TN - 1
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][TNA - 1][r] = acc[i][TN - 1][r];
          acc[i][TN - 1][r] = 0.0f;
        }
    }
  }
  if constexpr (FOLD) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = tot[i][j][r] + acc[i][j][r];
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  The residual is loaded for a whole
  // 32x32 tile at once from clamped (always valid) addresses: a per-element "load or not" makes the compiler branch
  // around every load and wait for each one.
  if constexpr (GRU) {
    static_assert(TM == 1 && TN == 3 && (WN == 2 || WN == 1) && W_NK && !PRO, "the GRU epilogue is written for waves of 32 x 96");
    const int C = s.cin;
    const int ch = 32 * WN * by + 32 * wn + l31;
    // r, z: one bias for the sum; n: the two halves apart
    const float br = g.bias[ch] + g.bias2[ch], bz = g.bias[C + ch] + g.bias2[C + ch];
    const float bni = g.bias[2 * C + ch], bnh = g.bias2[2 * C + ch];
    // h' and h by buffer operations from the rows' byte offsets (h and h' have C columns: the same offsets): rows of h
    // past its end read as zeros (the padding of lm:59-60), rows past M are not stored — both by the range check
    const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc((void*)g.out, 0, (int)(g.M * g.ld_out * 4), RSRC3);
    const int prow0 = wm * 32;
    unsigned voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) voff[r] = (unsigned)Rid[prow0 + (r & 3) + 8 * (r >> 2) + 4 * half] + 4u * (unsigned)ch;
    float hv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA1, voff[r], 0, 0));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // the cell's arithmetic is common.h's tln_gru_cell_value — the SAME function the small-lattice gates kernel calls
      // (accurate exp / tanh / division; round 3 had hardware approximations here).  Every multiply-add in it is spelled
      // out as ONE fused operation: left to the compiler's contraction, the single-cell and the batched instantiation of
      // this body (k_gather_gemm_v2_gru / _gru_multi) fused different ones and differed in the last bit — a lock-step
      // group must compute exactly what its sequences compute alone
      // tile 3: gi_n (x's share, parked), tile 2: gh_n
      const float hn = tln_gru_cell_value(acc[0][0][r] + br, acc[0][1][r] + bz, acc[0][3][r] + bni, acc[0][2][r] + bnh, hv[r]);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), rOut, voff[r], 0, 0);
    }
    return;
  }
  const bool has_res = g.res != nullptr;
  const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc((void*)g.out, 0, (int)(g.M * g.ld_out * 4), RSRC3);
  const __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? g.res : g.out), 0, (int)(g.M * g.ld_out * 4), RSRC3);
  const bool partial_cols = g.N % BN != 0;   // (uniform; only then a lane can sit on a column past N)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 32 * TN + j * 32 + l31;
      const bool ncol = n < g.N;
      const int nc = ncol ? n : g.N - 1;
      const float bias = g.bias ? g.bias[nc] : 0.f;
      const int prow0 = wm * 32 * TM + i * 32;           // block position of the tile's first row
      const int64_t mrow0 = m0 + prow0;
      unsigned voff[16];                                 // byte offsets of the 16 outputs (past the end: out of range)
#pragma unroll
      for (int r = 0; r < 16; ++r) voff[r] = (unsigned)Rid[prow0 + (r & 3) + 8 * (r >> 2) + 4 * half] + 4u * (unsigned)n;
      if (partial_cols) {   // (uniform: skipped by every product whose N is a multiple of the block's columns)
        if (!ncol) {
#pragma unroll
          for (int r = 0; r < 16; ++r) voff[r] = 0x80000000u;
        }
      }
      float rv[16];
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rRes, voff[r], 0, 0));
      }
      // (residual and ReLU under uniform branches around whole loops: inside one loop the compiler computes both
      //  alternatives and selects per element)
      float vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vv[r] = acc[i][j][r] + bias;
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] += rv[r];
      }
      if (g.relu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = fmaxf(vv[r], 0.f);
      }
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = (int)voff[r] >= 0;
        const float v = vv[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rOut, voff[r], 0, 0);
        const double dv = (double)(ok ? v : 0.0f);   // (one select on the float, not two on the double)
        s1 += dv;
        s2 += dv * dv;
      }
      if (g.stats) {  // wave-uniform
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (half == 0 && ncol && mrow0 < g.M) g.stats[(mrow0 >> 5) * g.N + n] = make_double2(s1, s2);
      }
    }
#endif
}

// Waves per SIMD a tile must keep for the workgroups per CU its LDS footprint allows (two-stage rings; LDS is granted in
// granules of 2 KB on this chip: TLN_V2_OCC=1 prints what the runtime fits): the 128 x 128 tile two workgroups of eight
// waves (4 per SIMD: 128 registers), the 64-row tile of 128 columns three of four waves (3 per SIMD: 168), the 128 x 64 tile
// two of eight.  Given to the compiler as the second launch bound: with the split accumulation's second accumulator set it
// would otherwise settle a few registers above those steps and lose a workgroup per CU.
template <int WM, int WN, int TM, int TN, int STAGES>
constexpr int v2_min_waves() {
#ifndef V2_N64_MIN_WAVES
#define V2_N64_MIN_WAVES 4   // (6 until round 4's last day: TLN_V2_OCC=1 showed TWO workgroups of the 128 x 64 tile per CU, not
#endif                       //  three — its 54 912 B round up to 27 LDS granules of 2 KB, 3 x 27 > 80 —, so the cap at 80 registers
                             //  bought four spilled lane constants and nothing else; measured equal, 1457 either way)
  if (STAGES == 2 && WM == 4 && WN == 2 && TM == 1 && TN == 1) return V2_N64_MIN_WAVES;
  if (STAGES == 2 && WM == 4 && WN == 2 && TM == 1 && TN == 2) return 4;
  if (STAGES == 2 && WM == 2 && WN == 2 && TM == 1 && TN == 2) return 3;
  return 1;
}

template <int WM, int WN, int TM, int TN, bool W_NK, bool PRO, int STAGES = V2_STAGES>
__global__ void __launch_bounds__(64 * WM * WN, (v2_min_waves<WM, WN, TM, TN, STAGES>())) k_gather_gemm_v2(const GemmArgs g) {
  v2_body<WM, WN, TM, TN, W_NK, PRO, false, STAGES>(g, (int)blockIdx.x, (int)blockIdx.y);
}

// Products of a shared launch <-> XCDs.  Workgroups are dealt round-robin over the eight XCDs in their linear order
// (observed; used for speed only), and every XCD has an L2 of its own: with blockIdx.z = product, the tiles of ONE product
// ran on all eight XCDs and each L2 saw the gathered rows of all eight sources (eight level-0 tensors: 61-188 MB through
// 4 MB).  With the product taken from the LOW bits of the linear block index, an XCD gathers from one source only
// (8 products; a pair of XCDs with 4, four with 2) — level 1 (4.5 MB) and level 2 then sit in its L2 — and the launch
// starts the heaviest tiles of ALL products first instead of product after product.
__device__ __forceinline__ void v2_multi_block(int xcd, int& bx, int& by, int& bz) {
  bx = (int)blockIdx.x;
  by = (int)blockIdx.y;
  bz = (int)blockIdx.z;
  const unsigned nz = gridDim.z;
  if (xcd && (nz == 8u || nz == 4u || nz == 2u)) {
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    bz = (int)(L & (nz - 1u));
    const unsigned r = L / nz;
    by = (int)(r / gridDim.x);
    bx = (int)(r - (unsigned)by * gridDim.x);
  }
}

// (V2_GRU_WAVES: the fused cell's register cap, as the second launch bound.  At its natural 140 (144 allocated, two waves per SIMD: 288 of a SIMD's 512)
//  a GRU workgroup cannot share a CU with another stream's 128-column workgroup since that one keeps a second accumulator
//  set (122 -> 128 allocated x 2 = 256); at 128 it can)
#ifndef V2_GRU_WAVES
#define V2_GRU_WAVES 4      // waves per SIMD the cell must allow: 4 = at most 128 VGPRs
#endif
template <int STAGES, int GWN = 2, int GWM = 4>
__global__ void __launch_bounds__(64 * GWM * GWN, (GWM == 4 ? V2_GRU_WAVES : 2)) k_gather_gemm_v2_gru(const GemmArgs g) {
  v2_body<GWM, GWN, 1, 3, true, false, true, STAGES>(g, (int)blockIdx.x, (int)blockIdx.y);
}

// the GRU cells of lock-stepped sequences in one launch (blockIdx.z = sequence; same weights, own x / h / out)
template <int STAGES, int GWN = 2, int GWM = 4>
__global__ void __launch_bounds__(64 * GWM * GWN, (GWM == 4 ? V2_GRU_WAVES : 2)) k_gather_gemm_v2_gru_multi(const GemmArgsN<TLN_GEMM_MULTI_MAX> gg) {
  int bx, by, bz;
  v2_multi_block(gg.xcd, bx, by, bz);
  const GemmArgs& g = gg.a[bz];
  if ((int64_t)bx * (32 * GWM) >= g.M) return;   // (the grid is sized for the largest lattice)
  v2_body<GWM, GWN, 1, 3, true, false, true, STAGES>(g, bx, by);
}

// several products of one shape class in one launch (blockIdx.z = product): the coarse levels of lock-stepped
// sequences, whose rows only together fill the chip with 128-row tiles
template <int WM, int WN, int TM, int TN, bool W_NK, bool PRO, int STAGES = V2_STAGES>
__global__ void __launch_bounds__(64 * WM * WN, (v2_min_waves<WM, WN, TM, TN, STAGES>())) k_gather_gemm_v2_multi(const GemmArgsN<TLN_GEMM_MULTI_MAX> gg) {
  int bx, by, bz;
  v2_multi_block(gg.xcd, bx, by, bz);
  const GemmArgs& g = gg.a[bz];
  if ((int64_t)(32 * TM * WM == 64 ? (bx >> 1) * 128 : bx * (32 * TM * WM)) >= g.M) return;   // (the grid is sized for the longest product)
  v2_body<WM, WN, TM, TN, W_NK, PRO, false, STAGES>(g, bx, by);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// below: gemm.hip's kernels.  Measured at M = 8.9k (level 1 of the headline lattice) with 64x64, 32x128, 64x128 block
// tiles of this kernel: 49-65 us against 52 us for the direct kernel on 128 -> 128 — too few chunks per block to pay
// for the ring's fill and the per-chunk barrier, and 1.1 tiles per SIMD leave no tile shape that balances.
// smallest M that takes this kernel: tln_options.v2_min_m, or the default (12288; env TLN_V2_MIN_M read once)
static int64_t v2_min_m(const tln_options& o) {
  static const int64_t def = getenv("TLN_V2_MIN_M") ? atoll(getenv("TLN_V2_MIN_M")) : 12288;
  return o.v2_min_m > 0 ? o.v2_min_m : def;
}

// the kernel addresses a source row and a weight row as a 32-bit byte offset into a buffer (the range check of the
// buffer delivers the zero rows): source and weights below 2 GiB each — 1.0M vertices x 256 channels still fit
static bool v2_bytes_ok(const GemmArgs& g) {
  const SrcDev& s = g.s[0];
  const int64_t lim = (1ll << 31) - 4096;
  if (g.res && g.ld_res != g.ld_out) return false;     // (the epilogue addresses residual and output by one offset)
  return s.src_rows * s.ld * 4 < lim && ((int64_t)s.taps * s.cin + g.N) * g.ldw * 4 < lim && g.M * g.ld_out * 4 < lim;
}

bool tln_gemm_v2_ok(const GemmArgs& g, bool w_is_nk, bool vec, const tln_options& o) {
  if ((o.v2_off & 1) || !vec || g.nsrc != 1 || g.M < v2_min_m(o)) return false;
  const SrcDev& s = g.s[0];
  if (s.cin % 32 != 0 || s.cin > 1024 || s.pad != 0.f) return false;
  if (!(s.taps == 1 || s.taps == TLN_TAPS)) return false;
  if (s.src_rows >= (1ll << 31) || g.M >= (1ll << 31)) return false;
  if (!v2_bytes_ok(g)) return false;
  // prologue: none, or GroupNorm affine + ReLU (scale/shift given, or partial sums with the scale/shift scratch)
  const bool affine = s.scale != nullptr || s.gn_part != nullptr;
  if (affine && (!s.relu || s.scale == nullptr || s.shift == nullptr)) return false;
  if (!affine && s.relu) return false;
  if (g.N % 32 != 0 && g.N != 96) return false;
  const int n = g.N;
  return n == 64 || n == 96 || n % 128 == 0 || n % 192 == 0;
}

// taps per partial sum of the split accumulation (v2_body): as many as make ~V2_FOLD_K k values (env TLN_V2_FOLD_K, read
// once; default 192: the measured noise level of the logits is then the fp32 CPU oracle's)
static int v2_fold_taps(int cin) {
  static const int target = getenv("TLN_V2_FOLD_K") ? atoi(getenv("TLN_V2_FOLD_K")) : 192;
  int t = (target + cin / 2) / (cin > 0 ? cin : 1);
  return t < 1 ? 1 : (t > TLN_TAPS ? TLN_TAPS : t);
}

// the row order of a product over a tap table (lattice.hip), unless switched off (TLN_V2_PERM_OFF, tln_options.v2_off bit 2)
static const int32_t* v2_perm_of(const GemmArgs& g, const tln_options& o) {
  static const bool off = getenv("TLN_V2_PERM_OFF") != nullptr;
  const SrcDev& s = g.s[0];
  if (off || (o.v2_off & 4) || s.table == nullptr || s.taps != TLN_TAPS) return nullptr;
  return tln_table_perm(s.table, g.M);
}

// which tiles run with a ring of TWO stages (bit 0: 128 x 128, bit 1: 128 x 64, bit 2: 128 x 192, bit 3: the GRU cell;
// TLN_V2_STAGES2, default all): the DMAs run one chunk ahead instead of two, but the workgroup needs 69 / 53 KB of LDS
// instead of 101 / 77 and two / three of them share a CU — measured +1.6 % clouds/s together.  The 128 x 192 tile and the
// GRU cell are alone on their CU either way (86 KB with two stages, 120 with three) as long as only their own launch
// runs — but with four streams in flight a 120 KB workgroup keeps every other stream's 53 / 69 KB workgroups off its CU
// and waits for a CU that is empty: with two stages they share (round 3: 1433 -> 1460 clouds/s at 4 x 8, nothing for a
// sequence alone; since the chunk barrier moved into the chunk's last step the DMAs of a two-stage ring still lead by
// a whole chunk)
static int v2_two_stage() {
  static const int v = getenv("TLN_V2_STAGES2") ? atoi(getenv("TLN_V2_STAGES2")) : 15;
  return v;
}

// (measurement: TLN_V2_XCD=0 keeps blockIdx.z = product in the shared launches, see v2_multi_block)
static int v2_xcd() {
  static const int on = (getenv("TLN_V2_XCD") != nullptr && atoi(getenv("TLN_V2_XCD")) == 0) ? 0 : 1;
  return on && tln_xcd_on();
}

template <int WM, int WN, int TM, int TN, int STAGES>
static size_t v2_lds_bytes(int cin) {
  constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int B_PIECES = (BN * 8 + NT - 1) / NT;
  return (size_t)STAGES * (BM * 128 + B_PIECES * NT * 16) + (size_t)BM * TLN_TAPS * 4 + (size_t)2 * cin * 4 + (size_t)(BM + 32) * 4;
}

template <int WM, int WN, int TM, int TN, bool W_NK, bool PRO, int STAGES = V2_STAGES>
static int launch_v2(GemmArgs& g, hipStream_t s, const tln_options& o) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  const size_t lds = v2_lds_bytes<WM, WN, TM, TN, STAGES>(g.s[0].cin);
  TLN_REQUIRE(lds <= 160 * 1024, "gemm v2: LDS %zu B", lds);
  auto kern = k_gather_gemm_v2<WM, WN, TM, TN, W_NK, PRO, STAGES>;
  static thread_local TlnLdsAttr attr;   // (one per template instantiation)
  TLN_HIP(tln_set_max_lds(attr, reinterpret_cast<const void*>(kern), (int)lds));
  dim3 grid((unsigned)(BM == 64 ? 2 * tln_cdiv(g.M, 128) : tln_cdiv(g.M, BM)), (unsigned)tln_cdiv(g.N, BN), 1);
  g.splits = 1;
  g.s[0].perm = v2_perm_of(g, o);
  g.fold_taps = v2_fold_taps(g.s[0].cin);
  g.s[0].order = g.s[0].perm ? tln_table_tile_order(g.s[0].table, g.M) : nullptr;
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, g);
  return TLN_OK;
}

static int v2_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;
  }
  return cus;
}

// 64-row tiles of four waves where a launch has few tiles per CU, 128-row tiles of eight where it has many.  A product of
// one sequence alone has 235 tiles of 128 rows for 256 CUs — one eight-wave workgroup per CU, its prologue and epilogue
// exposed, 21 CUs idle; as 470 tiles of 64 rows two to four independent workgroups share a CU (LDS: 36 / 53 / 69 KB for
// 64 / 128 / 192 columns), each a barrier domain of four waves.  The price is the weight tile staged once per 64 rows
// instead of 128 — L2 -> LDS traffic, of which there is enough.  MEASURED (round 4, same box): one sequence alone 664 ->
// 672 (64 columns) -> 678 (+ 128) -> 681 clouds/s (+ 192); four streams x 8: 64 columns 1457 -> 1462, 192 columns
// 1484 -> 1487 with the replay of a group's products 0.740 -> 0.748 of the peak.  Launches with many tiles per CU (the
// 1M-vertex lattices of SURVEY 8d's config 5) keep the 128-row tiles.  TLN_V2_ROWS64: 1 always, 0 never (measurements).
static bool v2_rows64(int64_t tiles128) {
  static const int env = getenv("TLN_V2_ROWS64") ? atoi(getenv("TLN_V2_ROWS64")) : -1;
  if (env >= 0) return env != 0;
  return tiles128 < (int64_t)16 * v2_cu_count();
}

template <bool W_NK, bool PRO>
static int dispatch_v2(GemmArgs& g, hipStream_t s, const tln_options& o) {
  const int n = g.N;
  // eight waves per block (two per SIMD: one issues MFMAs while the other waits for LDS or the barrier) measured 3-8 %
  // faster than four waves of twice the tile on every shape of the workload (TLN_V2_WAVES=4 brings those back)
  static const int waves = getenv("TLN_V2_WAVES") ? atoi(getenv("TLN_V2_WAVES")) : 8;
  if (waves == 8) {
    if (v2_rows64(tln_cdiv(g.M, 128) * tln_cdiv(n, n % 192 == 0 ? 192 : (n % 128 == 0 ? 128 : 64)))) {
      if (n % 192 == 0) return launch_v2<2, 2, 1, 3, W_NK, PRO, 2>(g, s, o);
      if (n % 128 == 0) return launch_v2<2, 2, 1, 2, W_NK, PRO, 2>(g, s, o);
      if (n == 64) return launch_v2<2, 2, 1, 1, W_NK, PRO, 2>(g, s, o);
    }
    if (n % 192 == 0)   // 128 x 192, 8 waves of 32 x 96
      return (v2_two_stage() & 4) ? launch_v2<4, 2, 1, 3, W_NK, PRO, 2>(g, s, o) : launch_v2<4, 2, 1, 3, W_NK, PRO>(g, s, o);
    if (n % 128 == 0) {   // 128 x 128, 8 waves of 32 x 64
      return (v2_two_stage() & 1) ? launch_v2<4, 2, 1, 2, W_NK, PRO, 2>(g, s, o) : launch_v2<4, 2, 1, 2, W_NK, PRO>(g, s, o);
    }
    if (n == 64)
      return (v2_two_stage() & 2) ? launch_v2<4, 2, 1, 1, W_NK, PRO, 2>(g, s, o) : launch_v2<4, 2, 1, 1, W_NK, PRO>(g, s, o);   // 128 x 64, 8 waves of 32 x 32 (also with the
                                                                          // GroupNorm prologue, which both column waves then apply: 37 against 38.6 us on 64 -> 64 x 9)
  }
  if (n % 192 == 0) return launch_v2<2, 2, 2, 3, W_NK, PRO>(g, s, o);   // 128 x 192, waves 64 x 96
  if (n % 128 == 0) return launch_v2<2, 2, 2, 2, W_NK, PRO>(g, s, o);   // 128 x 128, waves 64 x 64
  if (n == 96) return launch_v2<4, 1, 1, 3, W_NK, PRO>(g, s, o);        // 128 x 96, waves 32 x 96
  return (v2_two_stage() & 2) ? launch_v2<4, 1, 1, 2, W_NK, PRO, 2>(g, s, o) : launch_v2<4, 1, 1, 2, W_NK, PRO>(g, s, o);   // 128 x 64, waves 32 x 64
}

static bool v2_shape_ok(const GemmArgs& g, bool vec, const tln_options& o) {
  if ((o.v2_off & 1) || !vec || g.nsrc != 1) return false;
  const SrcDev& s = g.s[0];
  if (s.cin % 32 != 0 || s.cin > 1024 || s.pad != 0.f) return false;
  if (!(s.taps == 1 || s.taps == TLN_TAPS)) return false;
  if (s.src_rows >= (1ll << 31) || g.M >= (1ll << 31)) return false;
  if (!v2_bytes_ok(g)) return false;
  const bool affine = s.scale != nullptr || s.gn_part != nullptr;
  if (affine && (!s.relu || s.scale == nullptr || s.shift == nullptr)) return false;
  if (!affine && s.relu) return false;
  const int n = g.N;
  return n == 64 || n == 96 || n % 128 == 0 || n % 192 == 0;
}

bool tln_gemm_v2_multi_ok(const GemmArgs* g, int n, bool w_is_nk, const bool* vec, const tln_options& o) {
  static const bool off = getenv("TLN_V2_MULTI_OFF") != nullptr;
  if (off || n < 2 || n > TLN_GEMM_MULTI_MAX) return false;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (!v2_shape_ok(g[i], vec[i], o) || g[i].M < 1024) return false;
    const SrcDev &a = g[i].s[0], &b = g[0].s[0];
    if (g[i].N != g[0].N || a.cin != b.cin || a.taps != b.taps || (a.table != nullptr) != (b.table != nullptr) ||
        ((a.scale != nullptr || a.gn_part != nullptr) != (b.scale != nullptr || b.gn_part != nullptr)))
      return false;
    total += g[i].M;
  }
  (void)w_is_nk;
  static const int64_t multi_min = getenv("TLN_V2_MULTI_MIN_M") ? atoll(getenv("TLN_V2_MULTI_MIN_M")) : 0;
  return total >= (multi_min > 0 ? multi_min : v2_min_m(o));
}


template <int WM, int WN, int TM, int TN, bool W_NK, bool PRO, int STAGES = V2_STAGES>
static int launch_v2_multi(GemmArgs* g, int n, hipStream_t s, const tln_options& o) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  const size_t lds = v2_lds_bytes<WM, WN, TM, TN, STAGES>(g[0].s[0].cin);
  TLN_REQUIRE(lds <= 160 * 1024, "gemm v2: LDS %zu B", lds);
  auto kern = k_gather_gemm_v2_multi<WM, WN, TM, TN, W_NK, PRO, STAGES>;
  static thread_local TlnLdsAttr attr;   // (one per template instantiation)
  TLN_HIP(tln_set_max_lds(attr, reinterpret_cast<const void*>(kern), (int)lds));
  GemmArgsN<TLN_GEMM_MULTI_MAX> gg;
  gg.xcd = v2_xcd();
  int64_t mmax = 0;
  for (int i = 0; i < TLN_GEMM_MULTI_MAX; ++i) {
    gg.a[i] = g[i < n ? i : 0];
    gg.a[i].splits = 1;
    gg.a[i].s[0].perm = v2_perm_of(gg.a[i], o);
    gg.a[i].fold_taps = v2_fold_taps(gg.a[i].s[0].cin);
    gg.a[i].s[0].order = gg.a[i].s[0].perm ? tln_table_tile_order(gg.a[i].s[0].table, gg.a[i].M) : nullptr;
    if (i < n && g[i].M > mmax) mmax = g[i].M;
  }
  dim3 grid((unsigned)(BM == 64 ? 2 * tln_cdiv(mmax, 128) : tln_cdiv(mmax, BM)), (unsigned)tln_cdiv(g[0].N, BN), (unsigned)n);
  {   // (measurement: TLN_V2_OCC=1 prints, once per tile class, how many workgroups of it the runtime fits on a CU)
    static const bool occ = getenv("TLN_V2_OCC") != nullptr;
    static thread_local bool said = false;
    if (occ && !said) {
      said = true;
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * WM * WN, lds);
      fprintf(stderr, "[v2 occupancy] multi<%d,%d,%d,%d,%d,%d,%d> cin %d: %zu B of LDS, %d workgroups per CU\n", WM, WN, TM, TN,
              (int)W_NK, (int)PRO, STAGES, g[0].s[0].cin, lds, nb);
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, gg);
  return TLN_OK;
}

template <bool W_NK, bool PRO>
static int dispatch_v2_multi(GemmArgs* g, int n, hipStream_t s, const tln_options& o) {
  const int nn = g[0].N;
  int64_t tiles128 = 0;
  for (int i = 0; i < n; ++i) tiles128 += tln_cdiv(g[i].M, 128);
  tiles128 *= tln_cdiv(nn, nn % 192 == 0 ? 192 : (nn % 128 == 0 ? 128 : 64));
  if (nn % 192 == 0) {
    if (v2_rows64(tiles128)) return launch_v2_multi<2, 2, 1, 3, W_NK, PRO, 2>(g, n, s, o);
    return (v2_two_stage() & 4) ? launch_v2_multi<4, 2, 1, 3, W_NK, PRO, 2>(g, n, s, o) : launch_v2_multi<4, 2, 1, 3, W_NK, PRO>(g, n, s, o);
  }
  if (nn % 128 == 0) {
    // Tile height against the quantisation of the launch: two workgroups of this tile share a CU, and a CU's time is
    // (workgroups it gets) x (rows per workgroup).  The lock-stepped level-1 products have 52k-72k rows together — 407 to
    // 565 tiles of 128 rows on 256 CUs, i.e. "2 or 3 per CU" — so where the 128-row count lands just above a multiple
    // of the CU count, 96-row tiles (six waves) finish the launch in 3 x 96 rows per CU instead of 3 x 128.
    if constexpr (!W_NK && PRO) {
      // MEASURED (round 3, 4 streams x 8): never 1314, by this model 1284, always 1212 clouds/s — a six-wave workgroup
      // costs far more than 3/4 of an eight-wave one (three waves per SIMD instead of four while two workgroups share the
      // CU), so the tile stays off unless asked for (TLN_V2_BM96: 0 never = default, 1 always, -1 model)
      static const int env = getenv("TLN_V2_BM96") ? atoi(getenv("TLN_V2_BM96")) : 0;
      int64_t w128 = 0, w96 = 0;
      for (int i = 0; i < n; ++i) {
        w128 += tln_cdiv(g[i].M, 128);
        w96 += tln_cdiv(g[i].M, 96);
      }
      const int64_t cb = nn / 128;
      const double t128 = (double)tln_cdiv(w128 * cb, v2_cu_count()) * 1.0;
      const double t96 = (double)tln_cdiv(w96 * cb, v2_cu_count()) * 0.78;   // 0.75 of the rows + the fixed cost per workgroup
      if ((v2_two_stage() & 1) && (env == 1 || (env != 0 && t96 < t128)))
        return launch_v2_multi<3, 2, 1, 2, W_NK, PRO, 2>(g, n, s, o);
      // 64-row tiles of four waves (2 x 2, the same 32 x 64 per wave, three workgroups per CU): half the quantum of a
      // CU's time.  Where the 128-row count lands just above a multiple of the CU count (532 tiles: "3 per CU" for 2.08)
      // the launch takes 5 half-rounds instead of 3 whole ones.  MEASURED (4 streams x 8): the replay of a group's
      // products alone 113.4 -> 114.9 TFLOP/s by this model, the timed mode unchanged (1452 either way: the other streams
      // fill the idle CUs of a last round already).  On by the model (TLN_V2_BM64 / tln_options.v2_off bits 8, 16: 1 /
      // bit 8 always, 0 / bit 16 never)
      static const int env64 = getenv("TLN_V2_BM64") ? atoi(getenv("TLN_V2_BM64")) : -1;
      const int mode64 = (o.v2_off & 8) ? 1 : ((o.v2_off & 16) ? 0 : env64);
      int64_t w64 = 0;
      for (int i = 0; i < n; ++i) w64 += tln_cdiv(g[i].M, 64);
      const double t64 = (double)tln_cdiv(w64 * cb, v2_cu_count()) * 0.53;
      if ((v2_two_stage() & 1) && (mode64 == 1 || (mode64 != 0 && t64 < t128)))
        return launch_v2_multi<2, 2, 1, 2, W_NK, PRO, 2>(g, n, s, o);
    }
    return (v2_two_stage() & 1) ? launch_v2_multi<4, 2, 1, 2, W_NK, PRO, 2>(g, n, s, o) : launch_v2_multi<4, 2, 1, 2, W_NK, PRO>(g, n, s, o);
  }
  if (nn == 64) {
    // (measurement: TLN_V2_N64_WAVES=4 — four waves of 32 x 64 instead of eight of 32 x 32: every A fragment transformed once
    //  and used by two MFMA columns, half the vector instructions per block)
    static const int w64 = getenv("TLN_V2_N64_WAVES") ? atoi(getenv("TLN_V2_N64_WAVES")) : 8;
    if (w64 == 4) return launch_v2_multi<4, 1, 1, 2, W_NK, PRO, 2>(g, n, s, o);
    if (v2_rows64(tiles128)) return launch_v2_multi<2, 2, 1, 1, W_NK, PRO, 2>(g, n, s, o);   // 64 x 64 tile of four waves: four workgroups per CU
    return (v2_two_stage() & 2) ? launch_v2_multi<4, 2, 1, 1, W_NK, PRO, 2>(g, n, s, o) : launch_v2_multi<4, 2, 1, 1, W_NK, PRO>(g, n, s, o);
  }
  if (nn == 96) return launch_v2_multi<4, 1, 1, 3, W_NK, PRO>(g, n, s, o);
  return (v2_two_stage() & 2) ? launch_v2_multi<4, 1, 1, 2, W_NK, PRO, 2>(g, n, s, o) : launch_v2_multi<4, 1, 1, 2, W_NK, PRO>(g, n, s, o);
}

int tln_gemm_v2_launch_multi(GemmArgs* g, int n, bool w_is_nk, hipStream_t s, const tln_options& o) {
  const bool pro = g[0].s[0].scale != nullptr;
  if (w_is_nk) return pro ? dispatch_v2_multi<true, true>(g, n, s, o) : dispatch_v2_multi<true, false>(g, n, s, o);
  return pro ? dispatch_v2_multi<false, true>(g, n, s, o) : dispatch_v2_multi<false, false>(g, n, s, o);
}

// (measurement: TLN_GRU_HALF=1 runs the cell on 128 x 96 tiles of four waves — 32 channels x 3 gates, two workgroups per
//  CU's LDS, one's epilogue beside the other's K loop — instead of 128 x 192 tiles of eight)
static bool v2_gru_half() {
  static const bool on = getenv("TLN_GRU_HALF") != nullptr && atoi(getenv("TLN_GRU_HALF")) != 0;
  return on;
}

// (measurement: TLN_GRU_ROWS64=1 always / 0 never / 2 by v2_rows64's rule: the cell on 64 x 192 tiles of four waves)
static bool v2_gru_rows64(int64_t tiles128) {
  static const int env = getenv("TLN_GRU_ROWS64") ? atoi(getenv("TLN_GRU_ROWS64")) : 0;
  return env == 1 || (env == 2 && v2_rows64(tiles128));
}

// h' = GRUCell(x, pad(h)) as one two-source product with the cell in its epilogue (see v2_body)
bool tln_gemm_v2_gru_ok(int64_t V, int64_t Vh, int C, const tln_options& o) {
  static const bool off = getenv("TLN_GRU_FUSED_OFF") != nullptr;
  return !off && !(o.v2_off & 1) && V >= v2_min_m(o) && V * C * 4 < (1ll << 31) - 4096 && Vh >= 1 && Vh <= V && C % 64 == 0 && C <= 1024;
}
static void v2_gru_args(GemmArgs& g, const float* d_x, const float* d_h, int64_t Vh, int64_t V, int C, const float* d_w_ih,
                        const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out) {
  g = GemmArgs{};
  g.M = V;
  g.N = 3 * C;
  g.K0 = 2 * C;
  g.nsrc = 2;
  g.s[0].src = d_x;
  g.s[0].src_rows = V;
  g.s[0].ld = C;
  g.s[0].cin = C;
  g.s[0].taps = 2;            // the kernel's "taps" are the two sources here
  static const bool hskip_off = getenv("TLN_GRU_HSKIP_OFF") != nullptr;   // (measurement: blocks past the hidden state keep their h chunks)
  g.fold_taps = hskip_off ? 0 : 1;
  g.s[1].src = d_h;
  g.s[1].src_rows = Vh;
  g.s[1].ld = C;
  g.s[1].cin = C;
  g.s[1].taps = 1;
  g.W = d_w_ih;
  g.W2 = d_w_hh;
  g.ldw = C;
  g.bias = d_b_ih;
  g.bias2 = d_b_hh;
  g.out = d_out;
  g.ld_out = C;
  g.splits = 1;
}

int tln_gemm_v2_launch_gru(const float* d_x, const float* d_h, int64_t Vh, int64_t V, int C, const float* d_w_ih,
                           const float* d_w_hh, const float* d_b_ih, const float* d_b_hh, float* d_out, hipStream_t s) {
  GemmArgs g;
  v2_gru_args(g, d_x, d_h, Vh, V, C, d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_out);
  constexpr int BM = 128, BN = 192;
  const bool two = (v2_two_stage() & 8) != 0;
  if (v2_gru_rows64(tln_cdiv(V, 128) * (C / 64))) {   // 64 x 192 tile of four waves, two workgroups per CU
    const size_t lds = (size_t)2 * (64 + BN) * 128 + (size_t)64 * TLN_TAPS * 4 + (size_t)2 * C * 4 + (size_t)(64 + 32) * 4;
    static thread_local TlnLdsAttr attrr;
    TLN_HIP(tln_set_max_lds(attrr, reinterpret_cast<const void*>(k_gather_gemm_v2_gru<2, 2, 2>), (int)lds));
    hipLaunchKernelGGL((k_gather_gemm_v2_gru<2, 2, 2>), dim3((unsigned)tln_cdiv(V, 64), (unsigned)(C / 64), 1), dim3(256), lds, s, g);
    return TLN_OK;
  }
  if (v2_gru_half()) {   // 128 x 96 tile of four waves, two workgroups per CU
    const size_t lds = (size_t)2 * (BM + 96) * 128 + (size_t)BM * TLN_TAPS * 4 + (size_t)2 * C * 4 + (size_t)(BM + 32) * 4;
    static thread_local TlnLdsAttr attrh;
    TLN_HIP(tln_set_max_lds(attrh, reinterpret_cast<const void*>(k_gather_gemm_v2_gru<2, 1>), (int)lds));
    hipLaunchKernelGGL((k_gather_gemm_v2_gru<2, 1>), dim3((unsigned)tln_cdiv(V, BM), (unsigned)(C / 32), 1), dim3(256), lds, s, g);
    return TLN_OK;
  }
  const size_t lds = (size_t)(two ? 2 : 3) * (BM + BN) * 128 + (size_t)BM * TLN_TAPS * 4 + (size_t)2 * C * 4 + (size_t)(BM + 32) * 4;
  dim3 grid((unsigned)tln_cdiv(V, BM), (unsigned)(3 * C / BN), 1);
  if (two) {
    static thread_local TlnLdsAttr attr2;
    TLN_HIP(tln_set_max_lds(attr2, reinterpret_cast<const void*>(k_gather_gemm_v2_gru<2>), (int)lds));
    hipLaunchKernelGGL(k_gather_gemm_v2_gru<2>, grid, dim3(512), lds, s, g);
  } else {
    static thread_local TlnLdsAttr attr;
    TLN_HIP(tln_set_max_lds(attr, reinterpret_cast<const void*>(k_gather_gemm_v2_gru<3>), (int)lds));
    hipLaunchKernelGGL(k_gather_gemm_v2_gru<3>, grid, dim3(512), lds, s, g);
  }
  return TLN_OK;
}

// the cells of n <= TLN_GEMM_MULTI_MAX sequences (same weights) in ONE launch
int tln_gemm_v2_launch_gru_multi(int n, const float* const* d_x, const float* const* d_h, const int64_t* Vh, const int64_t* V,
                                 int C, const float* d_w_ih, const float* d_w_hh, const float* d_b_ih, const float* d_b_hh,
                                 float* const* d_out, hipStream_t s) {
  GemmArgsN<TLN_GEMM_MULTI_MAX> gg;
  gg.xcd = v2_xcd();
  int64_t vmax = 0;
  for (int i = 0; i < TLN_GEMM_MULTI_MAX; ++i) {
    const int k = i < n ? i : 0;
    v2_gru_args(gg.a[i], d_x[k], d_h[k], Vh[k], V[k], C, d_w_ih, d_w_hh, d_b_ih, d_b_hh, d_out[k]);
    if (i < n && V[k] > vmax) vmax = V[k];
  }
  constexpr int BM = 128, BN = 192;
  const bool two = (v2_two_stage() & 8) != 0;
  int64_t tiles128 = 0;
  for (int i = 0; i < n; ++i) tiles128 += tln_cdiv(V[i], 128) * (C / 64);
  if (v2_gru_rows64(tiles128)) {
    const size_t lds = (size_t)2 * (64 + BN) * 128 + (size_t)64 * TLN_TAPS * 4 + (size_t)2 * C * 4 + (size_t)(64 + 32) * 4;
    static thread_local TlnLdsAttr attrr;
    TLN_HIP(tln_set_max_lds(attrr, reinterpret_cast<const void*>(k_gather_gemm_v2_gru_multi<2, 2, 2>), (int)lds));
    hipLaunchKernelGGL((k_gather_gemm_v2_gru_multi<2, 2, 2>), dim3((unsigned)tln_cdiv(vmax, 64), (unsigned)(C / 64), (unsigned)n), dim3(256), lds, s, gg);
    return TLN_OK;
  }
  if (v2_gru_half()) {
    const size_t lds = (size_t)2 * (BM + 96) * 128 + (size_t)BM * TLN_TAPS * 4 + (size_t)2 * C * 4 + (size_t)(BM + 32) * 4;
    static thread_local TlnLdsAttr attrh;
    TLN_HIP(tln_set_max_lds(attrh, reinterpret_cast<const void*>(k_gather_gemm_v2_gru_multi<2, 1>), (int)lds));
    hipLaunchKernelGGL((k_gather_gemm_v2_gru_multi<2, 1>), dim3((unsigned)tln_cdiv(vmax, BM), (unsigned)(C / 32), (unsigned)n), dim3(256), lds, s, gg);
    return TLN_OK;
  }
  const size_t lds = (size_t)(two ? 2 : 3) * (BM + BN) * 128 + (size_t)BM * TLN_TAPS * 4 + (size_t)2 * C * 4 + (size_t)(BM + 32) * 4;
  dim3 grid((unsigned)tln_cdiv(vmax, BM), (unsigned)(3 * C / BN), (unsigned)n);
  if (two) {
    static thread_local TlnLdsAttr attr2;
    TLN_HIP(tln_set_max_lds(attr2, reinterpret_cast<const void*>(k_gather_gemm_v2_gru_multi<2>), (int)lds));
    hipLaunchKernelGGL(k_gather_gemm_v2_gru_multi<2>, grid, dim3(512), lds, s, gg);
  } else {
    static thread_local TlnLdsAttr attr;
    TLN_HIP(tln_set_max_lds(attr, reinterpret_cast<const void*>(k_gather_gemm_v2_gru_multi<3>), (int)lds));
    hipLaunchKernelGGL(k_gather_gemm_v2_gru_multi<3>, grid, dim3(512), lds, s, gg);
  }
  return TLN_OK;
}

int tln_gemm_v2_launch(GemmArgs& g, bool w_is_nk, hipStream_t s, const tln_options& o) {
  const bool pro = g.s[0].scale != nullptr;
  if (w_is_nk) return pro ? dispatch_v2<true, true>(g, s, o) : dispatch_v2<true, false>(g, s, o);
  return pro ? dispatch_v2<false, true>(g, s, o) : dispatch_v2<false, false>(g, s, o);
}
