#!/usr/bin/env python3
"""Micro-benchmark of k_gather_gemm on the shapes of the bench workload (interleaved variants in ONE process).
Each measurement is 20 back-to-back launches between two HIP events (the Python launch path costs ~8 us per call,
so very short kernels read host-bound here; use rocprofv3 for their true duration)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import _lib, ops                      # noqa: E402
from temporal_latticenet_amd.lattice import Lattice                # noqa: E402
from temporal_latticenet_amd.synthetic import make_sequence        # noqa: E402
from temporal_latticenet_amd import options as OPT  # noqa: E402

OPT.push()   # kernel-selection options of this host thread (tln_options; the library has no process-wide switch)


def main():
    seq = make_sequence(120000, 4, seed=1234)
    lat = Lattice.from_params([0.6] * 3, 1 << 18)
    for t, (p, v) in enumerate(seq):
        lat.distribute(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda(), reset_hashmap=(t == 0))
    l1 = lat.coarsen()
    l2 = l1.coarsen()
    levels = [lat, l1, l2]
    print("V:", [l.nr_lattice_vertices() for l in levels])
    shapes = [  # (level, cin, cout, taps, nk)
        (0, 128, 64, 9, False), (0, 64, 64, 9, False), (0, 192, 192, 9, False), (1, 128, 128, 9, False),
        (2, 64, 64, 9, False), (0, 192, 576, 1, True), (0, 192, 26, 1, True), (2, 256, 64, 1, True),
    ]
    # (splits, wm, groups); (0,0,0) = library heuristic
    variants = [(0, 0, 0), (1, 2, 4), (1, 1, 4), (2, 1, 2), (4, 1, 1), (4, 1, 2), (8, 1, 1), (2, 2, 2), (4, 2, 1), (4, 2, 2)]
    lib = _lib.lib()
    for lvl, cin, cout, taps, nk in shapes:
        L = levels[lvl]
        V = L.nr_lattice_vertices()
        x = torch.randn(V, cin, device="cuda")
        W = torch.randn((cout, taps * cin) if nk else (taps * cin, cout), device="cuda")
        tbl = L.neighbour_table_ptr() if taps == 9 else None
        res = {}
        for rnd in range(3):
            for sp, wm, g in variants:
                OPT.set(gemm_splits=sp, gemm_wm=wm)
                OPT.set(gemm_groups=g)
                src = ops.gemm_src(x, tbl, taps)
                out = torch.empty(V, cout, device="cuda")
                for _ in range(3):
                    ops.gather_gemm(V, W, src, w_is_nk=nk, out=out)
                torch.cuda.synchronize()
                # 20 launches captured in a hipGraph: the replay has no Python between kernels
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(20):
                        ops.gather_gemm(V, W, src, w_is_nk=nk, out=out)
                graph.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((sp, wm, g), []).append(e0.elapsed_time(e1) / 20 * 1e3)
        OPT.set(gemm_splits=0, gemm_wm=0)
        OPT.set(gemm_groups=0)
        fl = 2.0 * V * taps * cin * cout
        line = "  ".join("s%dw%dg%d:%5.1f" % (k[0], k[1], k[2], min(v)) for k, v in res.items())
        best = min(res.items(), key=lambda kv: min(kv[1]))
        print("M=%5d cin=%3d cout=%3d taps=%d | %s | best %s %.1f TF" % (V, cin, cout, taps, line, best[0],
                                                                    fl / min(best[1]) / 1e6))


if __name__ == "__main__":
    main()
