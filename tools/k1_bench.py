#!/usr/bin/env python3
"""K1 (distribute) alone: the frames of G lock-stepped sequences through one batch of launches
(tln_distribute_begin_multi) against G single calls, per frame of a 4-frame sequence of 120k-point scans.
  python tools/k1_bench.py [G=8] [reps=10] [batched|single|both]        env TLN_BK_ROWS / TLN_BK_PPB: bucket geometry overrides
Prints microseconds per frame (amortised over the group) and the algorithmic-bytes rate (128 N bytes per frame)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from temporal_latticenet_amd.lattice import Lattice  # noqa: E402
from temporal_latticenet_amd.synthetic import make_sequence  # noqa: E402
from temporal_latticenet_amd.workload import turned  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mode = sys.argv[3] if len(sys.argv) > 3 else "both"       # batched | single | both
    N, T = 120000, 4
    drive = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(N, T)]
    seqs = [turned(drive, j) for j in range(G)]
    lats = [Lattice.from_params([0.6] * 3, 1 << 18) for _ in range(G)]

    import ctypes as C
    from temporal_latticenet_amd import _lib
    from temporal_latticenet_amd.lattice import stream_ptr
    lib = _lib.lib()
    idx = [torch.empty(4 * N, dtype=torch.int32, device="cuda") for _ in range(G)]
    wts = [torch.empty(4 * N, dtype=torch.float32, device="cuda") for _ in range(G)]
    hs = (C.c_void_p * G)(*[l._h for l in lats])

    # as the frame program calls K1: no [4N, 5] rows (the pool reads the bins)
    def batch(t):
        if t == 0:
            _lib.check(lib.tln_lattice_clear_multi(hs, G, stream_ptr()), "clear")
        calls = (_lib.DistributeCall * G)()
        for k in range(G):
            c = calls[k]
            c.l, c.d_positions, c.d_values, c.n, c.val_dim = lats[k]._h, seqs[k][t][0].data_ptr(), seqs[k][t][1].data_ptr(), N, 1
            c.subtract_mean, c.d_distributed, c.d_indices, c.d_weights = 1, None, idx[k].data_ptr(), wts[k].data_ptr()
        _lib.check(lib.tln_distribute_begin_multi(calls, G, stream_ptr()), "begin_multi")
        for k in range(G):
            _lib.check(lib.tln_distribute_finish(lats[k]._h, stream_ptr()), "finish")

    def single(t):
        for k in range(G):
            if t == 0:
                _lib.check(lib.tln_lattice_clear(lats[k]._h, stream_ptr()), "clear")
            _lib.check(lib.tln_distribute(lats[k]._h, seqs[k][t][0].data_ptr(), seqs[k][t][1].data_ptr(), N, 1, 1, None,
                                          idx[k].data_ptr(), wts[k].data_ptr(), stream_ptr()), "distribute")

    for name, fn in (("batched", batch), ("single ", single)):
        if mode != "both" and mode != name.strip():
            continue
        for t in range(T):
            fn(t)
        torch.cuda.synchronize()
        tot = [0.0] * T
        for _ in range(reps):
            for t in range(T):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(t)
                e1.record()
                e1.synchronize()
                tot[t] += e0.elapsed_time(e1)
        us = [x * 1e3 / reps / G for x in tot]
        print("%s G=%d: us per frame (frames 0..3, incl. the clear on frame 0; no [4N,5] rows, as the frame program calls it): %s  mean of 1..3 %.1f us = %.0f GB/s"
              % (name, G, " ".join("%.1f" % u for u in us), sum(us[1:]) / 3, 128.0 * N / (sum(us[1:]) / 3 * 1e-6) / 1e9))


if __name__ == "__main__":
    main()
