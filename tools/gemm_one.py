#!/usr/bin/env python3
"""one conv shape, 20 launches (for rocprofv3 --pmc on k_gather_gemm)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import ops
from temporal_latticenet_amd.lattice import Lattice
from temporal_latticenet_amd.synthetic import make_sequence
cin, cout = int(sys.argv[1]), int(sys.argv[2])
seq = make_sequence(120000, 4, seed=1234)
lat = Lattice.from_params([0.6] * 3, 1 << 18)
for t, (p, v) in enumerate(seq):
    lat.distribute(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda(), reset_hashmap=(t == 0))
V = lat.nr_lattice_vertices()
x = torch.randn(V, cin, device="cuda"); W = torch.randn(9 * cin, cout, device="cuda"); out = torch.empty(V, cout, device="cuda")
for _ in range(20):
    ops.gather_gemm(V, W, ops.gemm_src(x, lat.neighbour_table_ptr(), 9), out=out)
torch.cuda.synchronize()
