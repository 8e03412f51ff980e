#!/usr/bin/env python3
"""K1 alone (distribute of the 4 frames of one calibrated 120k-point sequence, repeated): run under
rocprofv3 --kernel-trace --stats to get the per-kernel durations of the variant selected by the environment
(TLN_K1_LEGACY, TLN_BK_PPB, TLN_BK_ROWS).   python tools/k1_probe.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporal_latticenet_amd.lattice import Lattice          # noqa: E402
from temporal_latticenet_amd.synthetic import make_sequence  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seq = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(120000, 4)]
lat = Lattice.from_params([0.6] * 3, 100000)
for _ in range(2):
    for t, (p, v) in enumerate(seq):
        lat.distribute(p, v, reset_hashmap=(t == 0))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    for t, (p, v) in enumerate(seq):
        lat.distribute(p, v, reset_hashmap=(t == 0))
torch.cuda.synchronize()
print("distribute (operator call, host included): %.1f us per frame, V = %d" % (
    (time.perf_counter() - t0) / (4 * reps) * 1e6, lat.nr_lattice_vertices()))
import ctypes
import numpy as np
from temporal_latticenet_amd import _lib
L = ctypes.CDLL(_lib.LIB_PATH)
if hasattr(L, "tln_debug_bk_prof"):
    buf = torch.zeros(2 * 8192 * 8, dtype=torch.int64, device="cuda")
    L.tln_debug_bk_prof(ctypes.c_void_p(buf.data_ptr()))
    for t, (p, v) in enumerate(seq):
        buf.zero_()
        lat.distribute(p, v, reset_hashmap=(t == 0))
        torch.cuda.synchronize()
        a = buf.cpu().numpy().reshape(2, 8192, 8)
        for k, name in ((0, "insert"), (1, "place")):
            x = a[k]
            x = x[x[:, 0] > 0]
            t0 = x[:, 0].min()
            nz = [i for i in range(8) if x[:, i].max() > 0]
            print("frame", t, name, "blocks", len(x), "| stamp: mean / max after the first block start, 10 ns ticks:",
                  ["%d: %.0f / %d" % (i, (x[:, i] - t0).mean(), (x[:, i] - t0).max()) for i in nz])
    L.tln_debug_bk_prof(ctypes.c_void_p(0))
