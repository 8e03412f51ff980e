#!/usr/bin/env python3
"""The gather-GEMM shapes of the calibrated headline workload (sigma 0.6: V0 ~ 30k, V1 ~ 9k, V2 ~ 2.4k), every tile
variant of the tiled kernel and the direct kernel, 10 launches in a hipGraph each.  GPU box only."""
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
    shapes = [  # (level, cin, cout, taps, nk, groupnorm prologue)
        (0, 192, 192, 9, False), (0, 64, 64, 9, False), (0, 128, 64, 9, False), (0, 192, 192, 1, True),
        (0, 192, 96, 1, True), (1, 128, 128, 9, False), (1, 256, 128, 9, False), (2, 64, 64, 9, False),
        (2, 64, 256, 1, True), (2, 256, 64, 1, True), (0, 192, 576, 1, True), (0, 128, 384, 1, True), (0, 64, 192, 1, True),
        (0, 256, 128, 9, False),
    ]
    only = os.environ.get("TLN_SHAPES")
    if only:
        shapes = [shapes[int(i)] for i in only.split(",")]
    # (tm, tn, groups, direct)
    variants = [("v2", 0, 0, 0, 0), ("v2+gn", 0, 0, 0, 0), ("old heur", 0, 0, 0, 0), ("64x64", 1, 1, 0, -1),
                ("direct", 0, 0, 0, 1), ("direct+gn", 0, 0, 0, 1)]
    keep = os.environ.get("TLN_VARIANTS")
    if keep:
        variants = [v for v in variants if v[0] in keep.split(",")]
    lib = _lib.lib()
    for lvl, cin, cout, taps, nk in shapes:
        L = levels[lvl]
        V = L.nr_lattice_vertices()
        x = torch.randn(V, cin, device="cuda")
        W = torch.randn((cout, taps * cin) if nk else (taps * cin, cout), device="cuda")
        tbl = L.neighbour_table_ptr() if taps == 9 else None
        res = {}
        for name, tm, tn, g, direct in variants:
            if tn == 2 and cout < 128:
                continue
            OPT.set(v2_off=0 if name.startswith("v2") else 1, v2_min_m=1)
            OPT.set(gemm_tn=tn)
            OPT.set(gemm_direct=direct)
            if tm or tn:
                OPT.set(gemm_splits=1, gemm_wm=2)
            if name.endswith("+gn"):
                sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
                src = ops.gemm_src(x, tbl, taps, scale=sc, shift=sh, relu=True)
            else:
                src = ops.gemm_src(x, tbl, taps)
            out = torch.empty(V, cout, device="cuda")
            try:
                for _ in range(2):
                    ops.gather_gemm(V, W, src, w_is_nk=nk, out=out)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(10):
                        ops.gather_gemm(V, W, src, w_is_nk=nk, out=out)
                graph.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) / 10 * 1e3
            except Exception as e:      # a variant the shape cannot take
                res[name] = float("nan")
            OPT.set(gemm_tn=0)
            OPT.set(gemm_direct=0)
            OPT.set(gemm_splits=0, gemm_wm=0)
            OPT.set(v2_off=0, v2_min_m=0)
        fl = 2.0 * V * taps * cin * cout
        print("M=%5d cin=%3d cout=%3d taps=%d | " % (V, cin, cout, taps) +
              "  ".join("%s %6.1f us %5.1f TF" % (k, v, fl / v / 1e6) for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
