"""One sequence, its frames in flight together: the latency path of a single GPU.

The reference steps the frames of a sequence one after the other (train_ln.py:160-239), and so does `LNN_SEQ.forward`
here: a 120k-point frame is ~65 dependent launches of 6-40 us, most of which fill a fraction of the 256 CUs.  But frame
t+1 depends on frame t at FOUR points only — the hidden states of the fusion modules (early, middle, bottleneck, late:
models.py:129-148, lm:53-66, 207-235) — and through the vertex numbering of the shared hash table (models.py:287-289).
That is the dependence structure the frame-sharded multi-GPU path cuts at (dist.FrameShardRunner: rank g owns frame g,
receives every hidden state right before the first op that reads it and sends the new one on right after the last op
that writes it).  `FramePipeline` runs the same cut on ONE device: one slot per frame — a HIP stream, a host thread, a
model replica sharing the parameter tensors, a lattice of its own — and

  * the "all-gather" of the vertex keys is a hand-over inside the process: every slot hashes its own frame on a scratch
    lattice (first-touch order) and publishes the keys with a stream event; slot g inserts the keys of the frames before
    its own (`Lattice.insert_keys`), which reproduces the sequential numbering on every level;
  * the hidden states travel as (tensor, event) pairs through a queue between neighbouring slots: the sender records
    the event behind the copy of the new state, the receiver's stream waits for it — no host synchronisation.

Slot g is on segment s of frame g while slot g-1 is already past it: the sequence takes one frame plus (T-1) times its
longest segment instead of T frames, and the launches of up to T frames share the machine.  The arithmetic of every
frame is the frame program's own (`engine.FrameProgram.run_frame_sharded`), so the outputs are bit for bit those of the
sequential route (tests/test_gpu_pipeline.py).
"""
import contextlib
import io
import queue
import threading

import torch

from .dist import FrameShardPlan, FrameShardRunner
from .streams import share_parameters

__all__ = ["FramePipeline"]

_WAIT_S = 120.0      # a slot that waits this long for its neighbour reports instead of hanging


class _KeyBoard:
    """keys of the frames of the sequences in flight: (sequence, frame) -> (keys tensor, event)"""

    def __init__(self):
        self._cv = threading.Condition()
        self._items = {}

    def put(self, seq, frame, keys, event):
        with self._cv:
            self._items[(seq, frame)] = (keys, event)
            self._cv.notify_all()

    def get(self, seq, frame):
        with self._cv:
            if not self._cv.wait_for(lambda: (seq, frame) in self._items, timeout=_WAIT_S):
                raise RuntimeError("frame pipeline: the keys of frame %d never arrived" % frame)
            return self._items[(seq, frame)]

    def drop(self, seq):
        with self._cv:
            for k in [k for k in self._items if k[0] == seq]:
                del self._items[k]

    def clear(self):
        with self._cv:
            self._items.clear()


class _Slot(FrameShardRunner):
    """the owner of frame `slot` of every sequence: FrameShardRunner with the process group replaced by queues"""

    def __init__(self, model, make_lattice, plan, stream, inbox, outbox, board):
        super().__init__(model, make_lattice, plan, group=None, via_host=False, use_program=True)
        self.stream, self.inbox, self.outbox, self.board = stream, inbox, outbox, board

    trace = None      # a list: (label, event) marks on this slot's stream (tools/pipe_run.py draws the wavefront)

    def _mark(self, label):
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self.stream)
            self.trace.append((label, ev))

    _cur_seq = None   # (run generation, sequence index) of the job in hand: stamps every message this slot sends

    def _send_state(self, sid, t):
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.outbox.put((self._cur_seq, sid, t, ev))
        self._mark("sent %d" % sid)

    def _recv_state(self, sid):
        try:
            seq, got, t, ev = self.inbox.get(timeout=_WAIT_S)
        except queue.Empty:
            raise RuntimeError("frame pipeline: hidden state %d of the previous frame never arrived" % sid)
        if seq != self._cur_seq:      # a message left over from a sequence that failed: never consumed silently
            raise RuntimeError("frame pipeline: hidden state %d belongs to sequence %r, this slot is on %r"
                               % (got, seq, self._cur_seq))
        if got != sid:
            raise RuntimeError("frame pipeline: expected hidden state %d, the previous frame sent %d" % (sid, got))
        self._mark("wants %d" % sid)
        self.stream.wait_event(ev)
        if t is not None:
            # the tensor was allocated on the SENDER's stream: tell the caching allocator that this stream reads it too,
            # or the block returns to the sender's pool when the last reference drops and the sender (already on its next
            # sequence) may overwrite it while this stream's copy is still pending
            t.record_stream(self.stream)
        self._mark("has %d" % sid)
        return t if (t is not None and t.numel()) else None

    def run(self, seq, frame_data):
        """this slot's frame of sequence `seq`: publish its keys, take in the keys of the frames before it, run"""
        f = self.plan.frames[0]
        pos, val = frame_data
        self._cur_seq = seq
        self._mark("start")
        if self.plan.next_rank is not None:          # (nobody needs the last frame's keys)
            self.scratch.distribute(pos, val, reset_hashmap=True, subtract_mean=False)
            k = self.scratch.keys()
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.board.put(seq, f, k, ev)
        all_keys = [None] * self.plan.nr_frames
        for g in range(f):
            k, ev = self.board.get(seq, g)
            self.stream.wait_event(ev)
            k.record_stream(self.stream)      # produced on slot g's stream, read by insert_keys on this one
            all_keys[g] = k
        self._mark("keys in")
        out = self.run_sequence({f: frame_data}, all_keys)
        self._mark("end")
        return out


class FramePipeline:
    """T slots for the T frames of a sequence.  `run(sequences)` pushes every sequence (a list of T (positions, values)
    frames resident on the device) through the slots and returns the last frame's outputs (logsoftmax, raw scores), one
    pair per sequence; several sequences follow one another through the slots like a systolic array."""

    def __init__(self, base_model, make_model, make_lattice, warm_sequence, nr_frames=None):
        T = len(warm_sequence) if nr_frames is None else nr_frames
        self.nr_frames = T
        self.models = [base_model]
        quiet = contextlib.redirect_stdout(io.StringIO())
        for _ in range(T - 1):
            with quiet, torch.no_grad():
                m = make_model()
                m.train(base_model.training)
                lat = make_lattice()
                for t, (p, v) in enumerate(warm_sequence):           # creates the lazily built parameters
                    k = min(p.shape[0], 4096)
                    m(lat, p[:k], v[:k] if v is not None else None, t != len(warm_sequence) - 1, False)
                m.reset_sequence()
            self.models.append(share_parameters(m, base_model))
        self.streams = [torch.cuda.Stream() for _ in range(T)]
        self.board = _KeyBoard()
        links = [queue.Queue() for _ in range(T - 1)]             # link g: slot g -> slot g + 1
        self.slots = [_Slot(self.models[g], make_lattice, FrameShardPlan(T, g, T), self.streams[g],
                            links[g - 1] if g > 0 else None, links[g] if g < T - 1 else None, self.board)
                      for g in range(T)]
        self._links = links
        self._gen = 0                                             # run generation: stamps board entries and link messages
        self._jobs = [queue.Queue() for _ in range(T)]
        self._done = queue.Queue()
        self._threads = [threading.Thread(target=self._loop, args=(g,), daemon=True) for g in range(T)]
        for t in self._threads:
            t.start()

    def close(self):
        for q in self._jobs:
            q.put(None)
        for t in self._threads:
            t.join()
        self._threads = []
        for s in self.slots:
            s.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _loop(self, g):
        torch.cuda.set_device(self.streams[g].device)
        while True:
            job = self._jobs[g].get()
            if job is None:
                return
            seq, frame, opt, gen = job
            try:
                from . import options as O
                with torch.no_grad(), torch.cuda.stream(self.streams[g]), O.inherit(opt, gen):
                    out = self.slots[g].run(seq, frame)
                self._done.put((g, seq, out, None))
            except BaseException as e:          # reported by run(); the other slots time out on their queues
                self._done.put((g, seq, None, e))

    def run(self, sequences, keep_outputs=True):
        """every sequence through the pipeline; returns [(logsoftmax, raw scores)] of the last frames (None each without
        keep_outputs).  Synchronises the slots' streams with the caller's before it returns."""
        T = self.nr_frames
        cur = torch.cuda.current_stream()
        start = torch.cuda.Event()
        start.record(cur)
        for s in self.streams:
            s.wait_event(start)                       # the inputs were produced on the caller's stream
        from . import options as O
        self._gen += 1
        gen = self._gen
        for i, sq in enumerate(sequences):
            assert len(sq) == T, "a sequence of %d frames on a pipeline of %d slots" % (len(sq), T)
            for g in range(T):
                self._jobs[g].put(((gen, i), sq[g], O.current(), O.generation()))
        outs = [None] * len(sequences)
        err = None
        for _ in range(T * len(sequences)):
            g, seq, out, e = self._done.get()
            if e is not None and err is None:
                err = e
            if g == T - 1:
                if keep_outputs and out is not None:
                    outs[seq[1]] = (out[0], out[1])
                self.board.drop(seq)
        for s in self.streams:
            ev = torch.cuda.Event()
            ev.record(s)
            cur.wait_event(ev)
        for o in outs:                        # allocated on the last slot's stream, used on the caller's from here on
            if o is not None:
                for t in o:
                    if torch.is_tensor(t):
                        t.record_stream(cur)
        for m in self.models:                 # (slot 0's model is the caller's: leave it at the start of a sequence)
            m.reset_sequence()
        if err is not None:
            # nothing of the failed run may meet a later one: the slots are idle here (every job reported), so what is
            # left on the board and in the links is stale (and a message that slipped through would fail the generation
            # check of _recv_state / be an unknown board key)
            self.board.clear()
            for q in self._links:
                while True:
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        break
            raise err
        return outs
