"""Multi-GPU layer: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on the MI355X
node; "gloo" for the CPU rehearsals in tests/).  The reference is single-process / single-GPU, so
nothing here mirrors reference code; it must only preserve the single-GPU sequential semantics.

Two ways to use N GPUs:

 * sequence sharding (default of bench.py): a sequence owns its lattice and hidden states (train_ln.py:236-239),
   so ranks take disjoint sequences and never exchange data — `shard_items`, `max_over_ranks`.

 * frame sharding: rank g owns frame-slot g of every sequence.  Vertex numbering must equal the sequential one,
   so the ranks all-gather the first-touch-ordered NEW KEYS of their frames (`all_gather_rows`) and insert the
   frames before theirs in frame order (`Lattice.insert_keys`); the recurrence is honoured by handing each fusion
   module's hidden state [V_s, C_s] from rank g to rank g+1 (`send_tensor` / `recv_tensor`, point-to-point: one
   xGMI link) right after it is produced, which turns a stream of sequences into a systolic pipeline.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["init_from_env", "shard_items", "max_over_ranks", "all_gather_rows", "send_tensor", "recv_tensor",
           "HandoffError", "reset_handoff_counters", "FrameShardPlan", "FrameShardRunner"]


def init_from_env(backend=None, device_index=None):
    """RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run; returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        reset_handoff_counters()
    return rank, world


def shard_items(n_items, rank, world):
    """items (sequences) owned by `rank`: strided so that every rank gets the same count +-1"""
    return list(range(rank, n_items, world))


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the bench's step time is the slowest rank's)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_rows(rows, group=None):
    """all-gather of 2-D tensors with a different number of rows per rank -> list (rank order).
    Used for the per-frame new-vertex keys [dV_g, 3] int32."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros((m,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:c] for o, c in zip(out, counts)]


class HandoffError(RuntimeError):
    """a point-to-point hand-off went wrong: the message that arrived is not the one the receiver was waiting for (the
    two sides walked their fusion slots differently), or nothing arrived in time"""


_MAGIC = 0x544C4E48            # "TLNH"
_HDR = 8                       # magic, message number on this (src -> dst) channel, slot id, ndim, 4 extents
_sent, _received = {}, {}      # (ranks of the group, peer) -> messages completed so far on that channel
HANDOFF_TIMEOUT_S = float(os.environ.get("TLN_HANDOFF_TIMEOUT_S", "120"))
# Steady state: a hidden state travels as ONE message, its payload — the receiver knows the shape (rows = vertex count of
# the state's level before the frame, known from the key exchange; columns from the program).  TLN_HANDOFF_DEBUG=1 (and
# every hand-off whose shape the receiver cannot know: the operator-level hook route) puts the numbered header in front
# again: RCCL matches messages by order alone, the header is what turns a mis-ordered walk into an error.
HANDOFF_DEBUG = os.environ.get("TLN_HANDOFF_DEBUG", "0") not in ("", "0")


def reset_handoff_counters():
    _sent.clear()
    _received.clear()


def _channel(group, peer):
    """key of a (group, peer) channel that survives the group object: its global ranks (id() of a collected group can be
    reused by the next one)"""
    try:
        ranks = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    except Exception:
        ranks = ("world",)
    return ranks, peer


def _is_nccl(group):
    try:
        return dist.get_backend(group) == "nccl"
    except Exception:
        return False


def _wait(work, what, timeout_s, group=None):
    """gloo: block the host until the message is through, at most timeout_s (HandoffError instead of a hang).
    nccl (RCCL): `wait()` only makes the CURRENT STREAM wait for the transfer — the host runs ahead, which is what lets
    rank g start its next sequence while rank g+1 still works on this one (a host-side wait would serialise the
    pipeline); there is no host timeout on that path, a peer that never arrives is the launcher's to time out."""
    import datetime
    if _is_nccl(group):
        work.wait()
        return
    try:
        ok = work.wait(datetime.timedelta(seconds=timeout_s)) if timeout_s and timeout_s > 0 else work.wait()
    except RuntimeError as e:                                   # gloo raises on timeout
        raise HandoffError("%s: %s" % (what, e)) from e
    if ok is False:
        raise HandoffError("%s: timed out after %.0f s" % (what, timeout_s))


def send_tensor(t, dst, tag=0, group=None, timeout_s=None, header=True):
    """[header +] payload.  The header carries the shape (for receivers that cannot know it), a per-channel MESSAGE NUMBER
    and the SLOT id the sender believes it is serving: the RCCL backend ignores tags, messages between two ranks are
    matched by order alone, so a receiver that walked its fusion slots differently would silently take the wrong state —
    with the header it sees a message it did not expect and raises HandoffError.  header=False (both sides must agree,
    see HANDOFF_DEBUG): the payload alone."""
    key = _channel(group, dst)
    seq = _sent.get(key, 0)
    tmo = HANDOFF_TIMEOUT_S if timeout_s is None else timeout_s
    if header or HANDOFF_DEBUG:
        hdr = torch.tensor([_MAGIC, seq, int(tag), t.dim()] + list(t.shape) + [0] * (4 - t.dim()), dtype=torch.int64,
                           device=t.device)
        _wait(dist.isend(hdr, dst, group=group, tag=tag), "send of header %d (slot %d) to rank %d" % (seq, tag, dst), tmo, group)
    if t.numel():
        _wait(dist.isend(t.contiguous(), dst, group=group, tag=tag),
              "send of payload %d (slot %d) to rank %d" % (seq, tag, dst), tmo, group)
    _sent[key] = seq + 1          # (counted once the message is through)


def recv_tensor(src, device, dtype=torch.float32, tag=0, group=None, timeout_s=None, shape=None):
    """shape=None: the message starts with the numbered header (checked; one host synchronisation to read it).
    shape given: the payload alone, straight into a tensor of that shape — no header, no host synchronisation (unless
    HANDOFF_DEBUG asks both sides for the header; it must then agree with `shape`)."""
    key = _channel(group, src)
    seq = _received.get(key, 0)
    tmo = HANDOFF_TIMEOUT_S if timeout_s is None else timeout_s
    if shape is None or HANDOFF_DEBUG:
        hdr = torch.zeros(_HDR, dtype=torch.int64, device=device)
        _wait(dist.irecv(hdr, src, group=group, tag=tag), "receive of header %d (slot %d) from rank %d" % (seq, tag, src), tmo, group)
        h = hdr.tolist()
        if h[0] != _MAGIC or h[1] != seq or h[2] != int(tag) or not 0 <= h[3] <= 4:
            raise HandoffError("hand-off out of step: expected message %d for slot %d from rank %d, got magic %#x message %d "
                               "slot %d ndim %d" % (seq, tag, src, h[0], h[1], h[2], h[3]))
        got = h[4:4 + h[3]]
        if shape is not None and list(shape) != got:
            raise HandoffError("hand-off out of step: slot %d from rank %d has shape %s, expected %s" % (tag, src, got, list(shape)))
        shape = got
    t = torch.empty(tuple(shape), dtype=dtype, device=device)
    if t.numel():
        _wait(dist.irecv(t, src, group=group, tag=tag), "receive of payload %d (slot %d) from rank %d" % (seq, tag, src), tmo, group)
    _received[key] = seq + 1
    return t


class FrameShardPlan:
    """Which frames of a T-frame sequence a rank owns, who precedes / follows it, and which ranks form its group.

    world <= T : one group, rank g owns the contiguous block of T/world frames [g*T/world, (g+1)*T/world)
    world >  T : world/T independent groups of T ranks, one frame each (groups take different sequences)
    """

    def __init__(self, nr_frames, rank, world):
        if world <= nr_frames:
            if nr_frames % world:
                raise ValueError("frames (%d) must be a multiple of the ranks (%d)" % (nr_frames, world))
            self.group_size, self.nr_groups = world, 1
        else:
            if world % nr_frames:
                raise ValueError("ranks (%d) must be a multiple of the frames (%d)" % (world, nr_frames))
            self.group_size, self.nr_groups = nr_frames, world // nr_frames
        self.nr_frames = nr_frames
        self.group = rank // self.group_size
        self.slot = rank % self.group_size
        per = nr_frames // self.group_size
        self.frames = list(range(self.slot * per, (self.slot + 1) * per))
        base = self.group * self.group_size
        self.group_ranks = list(range(base, base + self.group_size))
        self.prev_rank = base + self.slot - 1 if self.slot > 0 else None
        self.next_rank = base + self.slot + 1 if self.slot < self.group_size - 1 else None

    def owns_last_frame(self):
        return self.frames[-1] == self.nr_frames - 1


class FrameShardRunner:
    """Runs a stream of T-frame sequences with the frames of each sequence sharded over the ranks of a group
    (FrameShardPlan) while keeping the single-GPU sequential semantics of LNN_SEQ (models.py:284-476):

      1. every rank hashes its own frames on a scratch lattice and the group all-gathers the first-touch-ordered
         keys of every frame (`all_gather_rows`); done for the whole batch of sequences up front so that the
         collective never sits between two pipeline stages;
      2. per sequence, a rank inserts the keys of the frames before its block (`Lattice.insert_keys`), which
         reproduces the sequential vertex numbering on every level, then runs its frames through the model;
      3. each fusion module receives its hidden state from the previous rank right before it is needed and
         sends it to the next rank right after it is produced (forward hooks, point-to-point), so rank g works on
         stage s of its frame while rank g-1 is already past it — over a stream of sequences this is a systolic
         pipeline in which every rank is busy with a different sequence.
         Ordering: between a pair of neighbouring ranks the messages are matched by ORDER alone (the NCCL / RCCL
         backend ignores tags): both sides walk the fusion slots in the model's fixed order early -> middle ->
         bottleneck -> late, one (header, payload) pair per slot and frame, and a rank only ever receives from its
         predecessor and sends to its successor — so the k-th send of rank g is the k-th receive of rank g+1.

    `via_host=True` stages tensors through host memory (gloo); otherwise tensors go GPU-to-GPU (RCCL over xGMI).
    """

    def __init__(self, model, make_lattice, plan, group=None, via_host=False, use_program=True):
        self.model, self.make_lattice, self.plan = model, make_lattice, plan
        self.group, self.via_host = group, via_host
        # use_program: drive the native frame program in segments (engine.FrameProgram.run_frame_sharded: the hidden
        # states enter and leave between ops); else, or when the model has no program (training mode, an unsupported
        # configuration), the operator-level route with the hand-off on forward hooks
        self.use_program = use_program
        self.scratch = make_lattice()
        self.lattice = make_lattice()
        self._recv_now = False
        self._send_now = False
        self._slots = []
        self._was_program = getattr(model, "use_frame_program", True)
        self._hooks = []
        self._hooked = False

    def _hook_route(self):
        """the operator-level route: hand-off on the fusion modules' forward hooks (until close())"""
        if not self._hooked:
            self.model.use_frame_program = False
            self._install_hooks()
            self._hooked = True

    def close(self):
        """removes the hand-off hooks and gives the model its frame program back"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._hooked = False
        self.model.use_frame_program = self._was_program

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- hidden-state hand-off ----------------------------------------------------------------
    def _fusion_modules(self):
        m = self.model
        mods = []
        if m.sequence_learning:
            if m.point_net_seq.fusion_module is not None:
                mods.append(m.point_net_seq.fusion_module)
            mods += [x for x in m.recurrent_fusion_modules if x is not None]
        return mods

    def _install_hooks(self):
        for slot, mod in enumerate(self._fusion_modules()):
            self._slots.append(mod)
            self._hooks.append(mod.register_forward_pre_hook(self._make_pre(slot)))
            self._hooks.append(mod.register_forward_hook(self._make_post(slot)))

    _shapes = None     # program route: {state id: (rows, cols)} of the states about to arrive (known before they do)

    def _recv_state(self, sid):
        """hidden state `sid` from the owner of the previous frame (None: it has none yet).  The transport: ranks of a
        process group here; pipeline.FramePipeline overrides both ends with in-process queues + stream events.
        On the program route the shape is known (rows = vertex count of the state's level before this frame, from the key
        exchange; columns from the program): the payload arrives alone, no header, no host synchronisation."""
        dev = "cpu" if self.via_host else "cuda"
        shape = self._shapes.get(sid) if self._shapes else None
        h = recv_tensor(self.plan.prev_rank, dev, tag=sid, group=self.group, shape=shape)
        return h.to("cuda", non_blocking=True) if h.numel() else None

    def _send_state(self, sid, t):
        """hidden state `sid` (a device tensor, or None) to the owner of the next frame; header-less exactly when the
        receiver knows the shape (program route on both sides: decided by the same flag on every rank)"""
        if t is None:
            t = torch.zeros(0, device="cuda")
        send_tensor(t.cpu() if self.via_host else t, self.plan.next_rank, tag=sid, group=self.group,
                    header=not (self._headerless and t.numel() > 0))

    _headerless = False

    def _make_pre(self, slot):
        def pre(mod, args):
            if self._recv_now and self.plan.prev_rank is not None:
                mod.h_lv = self._recv_state(slot)
            return None
        return pre

    def _make_post(self, slot):
        def post(mod, args, out):
            if self._send_now and self.plan.next_rank is not None:
                self._send_state(slot, mod.h_lv)
            return None
        return post

    # ---- key exchange ---------------------------------------------------------------------------
    def exchange_keys(self, sequences):
        """sequences: list over sequences of {frame index: (positions, values)} for the frames this rank owns.
        Returns, per sequence, the list of key tensors of ALL frames (frame order)."""
        per_seq = []
        for frames in sequences:
            mine = []
            for f in self.plan.frames:
                pos, val = frames[f]
                self.scratch.distribute(pos, val, reset_hashmap=True, subtract_mean=False)
                mine.append(self.scratch.keys())
            gathered = []     # frame-major
            for j in range(len(self.plan.frames)):
                k = mine[j].cpu() if self.via_host else mine[j]
                gathered.append(all_gather_rows(k, group=self.group))
            allk = [None] * self.plan.nr_frames
            per = len(self.plan.frames)
            for slot in range(self.plan.group_size):
                for j in range(per):
                    allk[slot * per + j] = gathered[j][slot]
            per_seq.append(allk)
        return per_seq

    # ---- a stream of sequences ------------------------------------------------------------------
    def run_stream(self, sequences, on_output=None):
        """sequences: list of {frame index: (positions, values)} (the frames this rank owns).  The key exchange is done
        PER SEQUENCE and one sequence ahead: the keys of sequence i + 1 are hashed (scratch lattice) and all-gathered
        before the frames of sequence i are enqueued — on a side stream when the tensors travel over RCCL, so that the
        collective overlaps sequence i's kernels — instead of one exchange for the whole batch up front (round 3), which
        put every rank's scratch K1 of the whole batch in front of the first frame.  Returns the outputs of
        run_sequence per sequence (or hands each to on_output and keeps nothing)."""
        outs = []
        if not sequences:
            return outs
        side = None
        if not self.via_host and torch.cuda.is_available():
            if getattr(self, "_key_stream", None) is None:
                self._key_stream = torch.cuda.Stream()
            side = self._key_stream

        def exchange(frames):
            if side is None:
                return self.exchange_keys([frames])[0], None
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)              # (the frames were produced on the caller's stream)
            with torch.cuda.stream(side):
                keys = self.exchange_keys([frames])[0]
                ev = torch.cuda.Event()
                ev.record(side)
            for k in keys:
                k.record_stream(cur)           # allocated on the side stream, consumed by insert_keys on the caller's
            return keys, ev

        nxt = exchange(sequences[0])
        for i, frames in enumerate(sequences):
            keys, ev = nxt
            nxt = exchange(sequences[i + 1]) if i + 1 < len(sequences) else None
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            out = self.run_sequence(frames, keys)
            if on_output is not None:
                on_output(i, out)
            else:
                outs.append(out)
        return outs

    # ---- one sequence ---------------------------------------------------------------------------
    def run_sequence(self, frames, all_keys):
        """frames: {frame index: (positions, values)} of the frames this rank owns.  Returns the model output
        of the last owned frame (logits on the rank that owns the final frame)."""
        model, lat = self.model, self.lattice
        model.reset_sequence()
        lat.clear()
        first = self.plan.frames[0]
        for f in range(first):
            k = all_keys[f]
            if k.shape[0]:
                lat.insert_keys(k.to("cuda") if k.device.type != "cuda" else k)
        prog = None
        if self.use_program and not self._hooked:
            model.first_sequence = True
            model.use_frame_program = True
            prog = model._program_for_this_frame(False)      # compiles / validates the program and resets its states
            if prog is None:
                self._hook_route()
        else:
            self._hook_route()
        out = None
        for j, f in enumerate(self.plan.frames):
            pos, val = frames[f]
            recv_now = (j == 0 and f > 0)
            send_now = (j == len(self.plan.frames) - 1 and f < self.plan.nr_frames - 1)
            early = f != self.plan.nr_frames - 1
            if prog is not None:
                out = self._program_frame(prog, lat, pos, val, f, early, recv_now, send_now)
                lat = out[2]
                continue
            if f > 0:
                model.first_sequence = False          # the lattice already holds the earlier frames (models.py:287-289)
            self._recv_now, self._send_now = recv_now, send_now
            out = model(lat, pos, val, early, False)
            self._recv_now = self._send_now = False
        return out

    def _program_frame(self, prog, lat, pos, val, f, early, recv_now, send_now):
        """one owned frame through the native frame program, the hidden states entering / leaving between its ops"""
        model = self.model
        expect = None
        if recv_now:
            # rows of the arriving states = vertex counts of their levels before this frame (prefix-stable numbering:
            # the keys of the earlier frames are already in, the coarse levels follow from them)
            expect = [lat.nr_lattice_vertices()]
            lvl = lat
            for _ in range(model.nr_downsamples):
                lvl = lvl.coarsen()
                expect.append(lvl.nr_lattice_vertices())

        # every state of a frame > 0 exists on the sender's side and has the row count of its level before this frame:
        # the messages of the program route need no header (both sides take this branch: same model, same flag)
        self._headerless = True
        self._shapes = None
        if recv_now:
            self._shapes = {}
            for sid in range(prog.nr_states):
                fr, lw, lvl = prog.state_ops(sid)
                self._shapes[sid] = (int(expect[lvl]), int(prog.state_cols(sid)))
        try:
            raw, lat = prog.run_frame_sharded(lat, pos, val, f == 0, early, self._recv_state if recv_now else None,
                                              self._send_state if send_now else None, expect)
        finally:
            self._shapes = None
            self._headerless = False
        model.first_sequence = False
        model._program_active = True
        if early and prog.stop_shape is not None:
            return raw, raw, lat
        fused = prog.take_logsm()
        return (fused if fused is not None else model.logsoftmax(raw)), raw, lat
