"""Several independent sequences at once on one GPU: one HIP stream, one host thread and one model replica per
sequence stream, all replicas sharing the same parameter tensors.

Why: a frame at sigma=0.6 touches a few thousand lattice vertices, so its ~75 launches are latency-bound and leave most
of the 256 CUs idle; sequences are independent units (a fresh lattice and fresh hidden states per sequence,
train_ln.py:236-239), so running a few of them concurrently fills the machine without any change to the arithmetic —
every stream computes exactly what it would compute alone (tests/test_gpu_streams.py).  The frame program
(engine.py) keeps the host cost per frame at two native calls, which release the GIL, so plain threads scale.
"""
import contextlib
import io
import os
import queue
import threading

# one hardware queue per stream: HIP's default of 4 is shared with the null stream, so two of four concurrent streams
# would serialise (measured: 1.9k vs 2.8k clouds/s).  Only effective if the HIP runtime has not initialised yet.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

__all__ = ["share_parameters", "SequenceStreams"]


def share_parameters(replica, base):
    """make every parameter / buffer of `replica` the SAME tensor object as in `base` (both already ran one sequence,
    so their lazily created parameters exist)"""
    base_params = dict(base.named_parameters())
    base_bufs = dict(base.named_buffers())
    mods = dict(replica.named_modules())
    for name, _ in list(replica.named_parameters()):
        owner, _, leaf = name.rpartition(".")
        if name not in base_params:
            raise KeyError("parameter %s exists in the replica only" % name)
        mods[owner]._parameters[leaf] = base_params[name]
    for name, _ in list(replica.named_buffers()):
        owner, _, leaf = name.rpartition(".")
        if name in base_bufs:
            mods[owner]._buffers[leaf] = base_bufs[name]
    return replica


class SequenceStreams:
    """S replicas of a model on S HIP streams.  `run(batches)` pushes batch i (a list of sequences, each a list of
    (positions, values) frames resident on the device) through replica i in its own thread and returns, per stream,
    the per-sequence outputs of the last frame."""

    def __init__(self, base_model, make_model, make_lattice, warm_sequence, n_streams, pairs=False):
        """pairs: True / 2..8: every stream steps that many sequences in lock-step (models.forward_group: their
        gather-GEMM launches are shared), i.e. pairs * n_streams sequences in flight -- the GPU runs at most four
        streams of a process at full rate"""
        self.group = 1 if not pairs else (2 if pairs is True else int(pairs))   # sequences per stream in lock-step
        assert 1 <= self.group <= 8
        self.pairs = self.group > 1
        self.n_streams = n_streams
        n_models = n_streams * self.group
        self.models = [base_model]
        self.make_lattice = make_lattice
        quiet = contextlib.redirect_stdout(io.StringIO())
        for _ in range(n_models - 1):
            with quiet, torch.no_grad():
                m = make_model()
                m.train(base_model.training)
                lat = make_lattice()
                for t, (p, v) in enumerate(warm_sequence):           # creates the lazily built parameters (their
                    k = min(p.shape[0], 4096)                        # shapes depend on channel counts only)
                    m(lat, p[:k], v[:k] if v is not None else None, t != len(warm_sequence) - 1, False)
                m.reset_sequence()
            self.models.append(share_parameters(m, base_model))
        self.streams = [torch.cuda.Stream() for _ in range(n_streams)]
        self._copy_streams = [torch.cuda.Stream() for _ in range(n_streams)]   # host -> device copies of the next frame
        self.lattices = [make_lattice() for _ in range(n_models)]
        # persistent workers (a fresh host thread pays HIP's per-thread set-up on its first call)
        self._jobs = [queue.Queue() for _ in range(n_streams)]
        self._done = queue.Queue()
        self._threads = [threading.Thread(target=self._loop, args=(i,), daemon=True) for i in range(1, n_streams)]
        for t in self._threads:
            t.start()

    def __len__(self):
        return self.n_streams

    def close(self):
        for i in range(1, self.n_streams):
            self._jobs[i].put(None)
        for t in self._threads:
            t.join()
        self._threads = []

    def _loop(self, i):
        torch.cuda.set_device(self.streams[i].device)
        while True:
            job = self._jobs[i].get()
            if job is None:
                return
            from . import options as O
            opt, gen, batch, keep = job
            with O.inherit(opt, gen):        # the kernel-selection options of the thread that called run()
                self._done.put((i,) + self._work(i, batch, keep))

    def _fetch(self, i, grp, t):
        """frame t of the sequences of a lock-step group: device tensors as they are; host tensors (pinned) start their
        way to the device on stream i's COPY stream.  Returns (positions, values, event or None)."""
        def on_device(x):
            return x is None or x.is_cuda          # (values may be None: a cloud without a value channel)

        if all(on_device(sq[t][0]) and on_device(sq[t][1]) for sq in grp):
            return [sq[t][0] for sq in grp], [sq[t][1] for sq in grp], None
        cs = self._copy_streams[i]
        with torch.cuda.stream(cs):
            # decided per tensor: only host tensors are copied, device tensors (and None) pass through unchanged
            ps = [x if on_device(x) else x.to("cuda", non_blocking=True) for x in (sq[t][0] for sq in grp)]
            vs = [x if on_device(x) else x.to("cuda", non_blocking=True) for x in (sq[t][1] for sq in grp)]
            ev = torch.cuda.Event()
            ev.record(cs)
        return ps, vs, ev

    def _arrived(self, i, fetched):
        ps, vs, ev = fetched
        if ev is not None:
            self.streams[i].wait_event(ev)
            for x in ps + vs:
                if x is not None:
                    x.record_stream(self.streams[i])    # allocated on the copy stream, used (and released) on this one
        return ps, vs

    def _work(self, i, sequences, keep_outputs):
        # (run() hands back last-frame outputs only: the early-return values of the other frames are not materialised)
        mine = self.models[self.group * i:self.group * i + self.group] if self.pairs else [self.models[i]]
        before = [m.keep_early_values for m in mine]
        for m in mine:
            m.keep_early_values = bool(os.environ.get("TLN_KEEP_EARLY"))      # (measurement: the copies back in)
        try:
            return self._work_inner(i, sequences, keep_outputs)
        finally:
            for m, b in zip(mine, before):
                m.keep_early_values = b

    def _work_inner(self, i, sequences, keep_outputs):
        try:
            outs = []
            with torch.no_grad(), torch.cuda.stream(self.streams[i]):
                if self.pairs:
                    from .models import forward_group
                    g = self.group
                    models, lats = self.models[g * i:g * i + g], self.lattices[g * i:g * i + g]
                    k = 0
                    while k + g <= len(sequences) and all(len(sequences[k + j]) == len(sequences[k]) for j in range(g)):
                        grp = sequences[k:k + g]
                        nxt = self._fetch(i, grp, 0)
                        for t in range(len(grp[0])):
                            ps, vs = self._arrived(i, nxt)
                            # frames waiting in (pinned) host memory: the next frame's copies run on the stream pool's copy
                            # stream while this frame computes
                            nxt = self._fetch(i, grp, t + 1) if t + 1 < len(grp[0]) else None
                            res = forward_group(models, lats, ps, vs, t != len(grp[0]) - 1)
                            lats = [r[2] for r in res]
                        for mod in models:
                            mod.reset_sequence()
                        if keep_outputs:
                            outs += [r[1] for r in res]
                        k += g
                    rest, model, lat = sequences[k:], models[0], lats[0]
                else:
                    rest, model, lat = sequences, self.models[i], self.lattices[i]
                for seq in rest:
                    nxt = self._fetch(i, [seq], 0)
                    for t in range(len(seq)):
                        (p,), (v,) = self._arrived(i, nxt)
                        nxt = self._fetch(i, [seq], t + 1) if t + 1 < len(seq) else None
                        out, raw, lat = model(lat, p, v, t != len(seq) - 1, False)
                    model.reset_sequence()
                    if keep_outputs:
                        outs.append(raw)
                self.streams[i].synchronize()
            return outs, None
        except BaseException as e:      # surfaced by run()
            return None, e

    def run(self, batches, keep_outputs=False):
        n = self.n_streams
        assert len(batches) == n
        cur = torch.cuda.current_stream()
        for s in self.streams:           # inputs prepared on the caller's stream are ready for every worker
            s.wait_stream(cur)
        from . import options as O
        for i in range(1, n):
            self._jobs[i].put((O.current(), O.generation(), batches[i], keep_outputs))
        results, errors = [None] * n, [None] * n
        results[0], errors[0] = self._work(0, batches[0], keep_outputs)
        for _ in range(1, n):
            i, outs, err = self._done.get()
            results[i], errors[i] = outs, err
        for e in errors:
            if e is not None:
                raise e
        for s in self.streams:
            cur.wait_stream(s)
        return results
