#!/usr/bin/env python3
"""k_gather_gemm in its large-M regime (a fine lattice: sigma 0.05 -> ~170k vertices)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import _lib, ops
from temporal_latticenet_amd.lattice import Lattice
from temporal_latticenet_amd.synthetic import make_sequence
from temporal_latticenet_amd import options as OPT  # noqa: E402

OPT.push()   # kernel-selection options of this host thread (tln_options; the library has no process-wide switch)

seq = make_sequence(120000, 4, seed=1234)
lat = Lattice.from_params([0.05] * 3, 1 << 19)
for t, (p, v) in enumerate(seq):
    lat.distribute(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda(), reset_hashmap=(t == 0))
V = lat.nr_lattice_vertices()
print("V =", V)
lib = _lib.lib()
for cin, cout, taps in [(192, 192, 9), (64, 64, 9), (128, 128, 9), (128, 256, 9), (192, 128, 9), (192, 576, 1)]:
    x = torch.randn(V, cin, device="cuda")
    W = torch.randn(taps * cin, cout, device="cuda")
    tbl = lat.neighbour_table_ptr() if taps == 9 else None
    out = torch.empty(V, cout, device="cuda")
    res = {}
    for tm, tn in [(0, 0), (1, 1), (2, 1), (1, 2), (2, 2)]:
        if tn == 2 and cout <= 64:
            continue
        OPT.set(gemm_tn=tn)
        if (tm, tn) != (0, 0):
            OPT.set(gemm_splits=1, gemm_wm=2)
        src = ops.gemm_src(x, tbl, taps)
        for _ in range(2):
            ops.gather_gemm(V, W, src, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gather_gemm(V, W, src, out=out)
        e1.record()
        torch.cuda.synchronize()
        res[(tm, tn)] = e0.elapsed_time(e1) / 5
        OPT.set(gemm_tn=0)
        OPT.set(gemm_splits=0, gemm_wm=0)
    fl = 2.0 * V * taps * cin * cout
    print("cin=%3d cout=%3d taps=%d | " % (cin, cout, taps) + "  ".join("t%d%d: %6.3f ms %5.1f TF" % (k[0], k[1], v, fl / v / 1e9) for k, v in res.items()))
