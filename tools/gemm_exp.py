#!/usr/bin/env python3
"""one heavy shape, direct kernel, G in (1, 2, 4): time per launch (for kernel experiments)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import _lib, ops
from temporal_latticenet_amd.lattice import Lattice
from temporal_latticenet_amd.synthetic import make_sequence
from temporal_latticenet_amd import options as OPT  # noqa: E402

OPT.push()   # kernel-selection options of this host thread (tln_options; the library has no process-wide switch)
cin, cout = int(sys.argv[1]), int(sys.argv[2])
seq = make_sequence(120000, 4, seed=1234)
lat = Lattice.from_params([0.6] * 3, 1 << 18)
for t, (p, v) in enumerate(seq):
    lat.distribute(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda(), reset_hashmap=(t == 0))
V = lat.nr_lattice_vertices()
x = torch.randn(V, cin, device="cuda"); W = torch.randn(9 * cin, cout, device="cuda"); out = torch.empty(V, cout, device="cuda")
lib = _lib.lib()
OPT.set(gemm_direct=1)
res = []
for G in (1, 2, 4, 8, 12):
    OPT.set(gemm_groups=G)
    src = ops.gemm_src(x, lat.neighbour_table_ptr(), 9)
    for _ in range(3):
        ops.gather_gemm(V, W, src, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.gather_gemm(V, W, src, out=out)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    res.append("G%d:%.1f" % (G, e0.elapsed_time(e1) / 20 * 1e3))
print("exp=%s M=%d %dx%d  %s" % (os.environ.get("TLN_EXP", "0"), V, cin, cout, "  ".join(res)))
