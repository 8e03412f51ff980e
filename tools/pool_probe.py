#!/usr/bin/env python3
"""K1 + K2 alone (distribute and PointNet pool of the 4 frames of one calibrated 120k-point sequence, repeated): for
rocprofv3 runs (--kernel-trace --stats, --pmc ...).   python tools/pool_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd import ops                      # noqa: E402
from temporal_latticenet_amd.lattice import Lattice          # noqa: E402
from temporal_latticenet_amd.synthetic import make_sequence  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
seq = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(120000, 4)]
lat = Lattice.from_params([0.6] * 3, 100000)
g = torch.Generator().manual_seed(5)
Ws = [(torch.randn(16, 4, generator=g) * 0.5).cuda(), (torch.randn(32, 16, generator=g) * 0.3).cuda(), (torch.randn(64, 32, generator=g) * 0.3).cuda()]
Bs = [(torch.randn(16, generator=g) * 0.1).cuda(), (torch.randn(32, generator=g) * 0.1).cuda(), (torch.randn(64, generator=g) * 0.1).cuda()]
for _ in range(reps):
    for t, (p, v) in enumerate(seq):
        d, i, w = lat.distribute(p, v, reset_hashmap=(t == 0))
        out = ops.pointnet_pool(lat, d, i, Ws, Bs, 4)
torch.cuda.synchronize()
print("V =", lat.nr_lattice_vertices(), "checksum %.6f" % float(out.double().sum()))
