#!/usr/bin/env python3
"""The K-heavy level-0/1 products of the headline workload (the five launches that take ~45% of a frame's GEMM time):
direct kernel over G against the LDS-tiled kernel over (splits, wm, groups).  20 launches per hipGraph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import _lib, ops                      # noqa: E402
from temporal_latticenet_amd.lattice import Lattice                # noqa: E402
from temporal_latticenet_amd.synthetic import make_sequence        # noqa: E402
from temporal_latticenet_amd import options as OPT  # noqa: E402

OPT.push()   # kernel-selection options of this host thread (tln_options; the library has no process-wide switch)


_side = None


def timed(fn):
    global _side
    if _side is None:
        _side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(_side):      # the split-K workspace is per stream: size it before the capture
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=_side):
        for _ in range(20):
            fn()
    graph.replay()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best


def main():
    seq = make_sequence(120000, 4, seed=1234)
    lat = Lattice.from_params([0.6] * 3, 1 << 18)
    for t, (p, v) in enumerate(seq):
        lat.distribute(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda(), reset_hashmap=(t == 0))
    l1 = lat.coarsen()
    levels = [lat, l1]
    lib = _lib.lib()
    shapes = [(0, 192, 192), (0, 256, 128), (1, 256, 128), (1, 128, 128), (0, 128, 64), (0, 64, 64)]
    only = sys.argv[1:]  # optional: variant filter
    for lvl, cin, cout in shapes:
        L = levels[lvl]
        V = L.nr_lattice_vertices()
        x = torch.randn(V, cin, device="cuda")
        W = torch.randn(9 * cin, cout, device="cuda")
        out = torch.empty(V, cout, device="cuda")
        src = ops.gemm_src(x, L.neighbour_table_ptr(), 9)
        fn = lambda: ops.gather_gemm(V, W, src, out=out)
        res = {}
        OPT.set(gemm_direct=0)
        res["auto"] = timed(fn)
        OPT.set(gemm_direct=1)
        for G in (2, 3, 4, 6, 9, 12):
            OPT.set(gemm_groups=G)
            res["d%d" % G] = timed(fn)
        OPT.set(gemm_groups=0)
        OPT.set(gemm_direct=-1)
        for sp, wm, g in [(1, 1, 4), (1, 1, 2), (1, 2, 4), (1, 2, 2), (2, 1, 2), (2, 2, 2), (2, 2, 4), (4, 2, 2)]:
            OPT.set(gemm_splits=sp, gemm_wm=wm)
            OPT.set(gemm_groups=g)
            res["t s%dw%dg%d" % (sp, wm, g)] = timed(fn)
        for tm, tn in [(2, 1), (1, 2), (2, 2)]:
            OPT.set(gemm_splits=0, gemm_wm=0)
            OPT.set(gemm_groups=0)
            OPT.set(gemm_tn=tn)
            res["t tm%dtn%d" % (tm, tn)] = timed(fn)
        OPT.set(gemm_tn=0)
        OPT.set(gemm_splits=0, gemm_wm=0)
        OPT.set(gemm_groups=0)
        OPT.set(gemm_direct=0)
        fl = 2.0 * V * 9 * cin * cout
        print("M=%5d cin=%3d cout=%3d | %s" % (V, cin, cout, "  ".join("%s:%.1f" % kv for kv in res.items())), flush=True)
        b = min(res.items(), key=lambda kv: kv[1])
        print("      best %s %.1f us %.1f TF" % (b[0], b[1], fl / b[1] / 1e6), flush=True)


if __name__ == "__main__":
    main()
