#!/usr/bin/env python3
"""Where does one gather-GEMM block spend its cycles?  s_memtime stamps of block (0,0,0)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import _lib, ops
from temporal_latticenet_amd.lattice import Lattice
from temporal_latticenet_amd.synthetic import make_sequence
from temporal_latticenet_amd import options as OPT  # noqa: E402

OPT.push()   # kernel-selection options of this host thread (tln_options; the library has no process-wide switch)
seq = make_sequence(120000, 4, seed=1234)
lat = Lattice.from_params([0.6] * 3, 1 << 18)
for t, (p, v) in enumerate(seq):
    lat.distribute(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda(), reset_hashmap=(t == 0))
lat.prepare_levels(2)
l1 = lat.coarsen(); l2 = l1.coarsen()
lib = _lib.lib()
buf = torch.zeros(8, dtype=torch.int64, device="cuda")
norm = torch.nn.GroupNorm(32, 64).cuda()
for name, L, cin, cout, taps, use_gn in [("conv 64->64 V0", lat, 64, 64, 9, True), ("conv 64->64 V0 noGN", lat, 64, 64, 9, False),
                                         ("conv 192->192 V0", lat, 192, 192, 9, True), ("lin 256->64 V2", l2, 256, 64, 1, False),
                                         ("conv 128->128 V1", l1, 128, 128, 9, True)]:
    V = L.nr_lattice_vertices()
    x = torch.randn(V, cin, device="cuda")
    prod = ops.gather_gemm(V, torch.randn(cin, cin, device="cuda"), ops.gemm_src(x), stats=True)   # gives partials
    W = torch.randn(taps * cin, cout, device="cuda")
    nrm = torch.nn.GroupNorm(32, cin).cuda()
    tbl = L.neighbour_table_ptr() if taps == 9 else None
    for _ in range(3):
        ops.gather_gemm(V, W, ops.gemm_src(prod, tbl, taps), stats=True, gn=(prod, nrm, True) if use_gn else None)
    OPT.set(gemm_stamps=C.c_void_p(buf.data_ptr()))
    ops.gather_gemm(V, W, ops.gemm_src(prod, tbl, taps), stats=True, gn=(prod, nrm, True) if use_gn else None)
    OPT.set(gemm_stamps=None)
    torch.cuda.synchronize()
    s = buf.cpu().tolist()
    d = [(s[i + 1] - s[i]) for i in range(4)]
    print("%-22s V=%5d | prologue %6d  K-loop %6d  reduce %6d  epilogue %6d  cycles (total %.1f us @2.4GHz... x? clock 100MHz ticks=%d)"
          % (name, V, d[0], d[1], d[2], d[3], (s[4] - s[0]) / 2400.0, s[4] - s[0]))
    if s[5]:
        print("    direct kernel: tap indices arrived +%d, first chunk loads issued +%d, GroupNorm partials summed +%d (cycles after start)" % (s[5] - s[0], s[6] - s[0], (s[7] - s[0]) if s[7] else 0))
    buf.zero_()
