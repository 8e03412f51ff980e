#!/usr/bin/env python3
"""distribute + PointNet pool only (for rocprofv3 --pmc on k_pool_chunks)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import ops
from temporal_latticenet_amd.lattice import Lattice
from temporal_latticenet_amd.synthetic import make_sequence
pos, val = make_sequence(120000, 1, seed=1234)[0]
lat = Lattice.from_params([0.6] * 3, 1 << 18)
d, i, w = lat.distribute(torch.from_numpy(pos).cuda(), torch.from_numpy(val).cuda())
g = torch.Generator().manual_seed(0)
Ws = [torch.randn(16, 4, generator=g).cuda(), torch.randn(32, 16, generator=g).cuda(), torch.randn(64, 32, generator=g).cuda()]
Bs = [torch.randn(16, generator=g).cuda(), torch.randn(32, generator=g).cuda(), torch.randn(64, generator=g).cuda()]
for _ in range(10):
    out = ops.pointnet_pool(lat, d, i, Ws, Bs, 4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.pointnet_pool(lat, d, i, Ws, Bs, 4)
e1.record(); torch.cuda.synchronize()
print("pool: %.1f us per call, V=%d" % (e0.elapsed_time(e1) / 20 * 1e3, lat.nr_lattice_vertices()))
