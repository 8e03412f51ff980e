#!/usr/bin/env python3
"""Where does a gemm_v2 workgroup's K loop spend its time?  Needs a library built with -DTLN_V2_STAMPS
(TLN_EXTRA_FLAGS=-DTLN_V2_STAMPS python -m temporal_latticenet_amd.build after touching gemm_v2.hip): every wave of a
sample of blocks adds its loop time, the time it waited for its own DMAs and the time it stood at the chunk barrier
(s_memtime ticks of 10 ns) to counters behind tln_gemm_debug_stamps."""
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
lib = _lib.lib()
buf = torch.zeros(32 + 60 * 16, dtype=torch.int64, device="cuda")
V = lat.nr_lattice_vertices()
for name, cin, cout, taps, use_gn in [("64->64 x9 gn", 64, 64, 9, True), ("128->64 x9", 128, 64, 9, False), ("192->192 x9 gn", 192, 192, 9, True),
                                      ("256->128 x9 gn", 256, 128, 9, True), ("128->384 x1", 128, 384, 1, False)]:
    x = torch.randn(V, cin, device="cuda")
    prod = ops.gather_gemm(V, torch.randn(cin, cin, device="cuda"), ops.gemm_src(x), stats=True)
    W = torch.randn(taps * cin, cout, device="cuda")
    nrm = torch.nn.GroupNorm(32, cin).cuda()
    tbl = lat.neighbour_table_ptr() if taps == 9 else None
    run = lambda: ops.gather_gemm(V, W, ops.gemm_src(prod, tbl, taps), stats=True, gn=(prod, nrm, True) if use_gn else None)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    buf.zero_()
    OPT.set(gemm_stamps=C.c_void_p(buf.data_ptr()))
    run()
    OPT.set(gemm_stamps=None)
    torch.cuda.synchronize()
    s = buf.cpu().tolist()
    tot, dma, bar, chunks, waves = s[16:21]
    if waves == 0:
        print("%-16s no stamps (library built without -DTLN_V2_STAMPS?)" % name)
        continue
    # block 37's timeline: per chunk the arrival of every wave at the barrier and the release
    import numpy as np
    tl = np.array(s[32:32 + 60 * 16], dtype=np.int64).reshape(60, 8, 2)
    nck = int((tl[:, 0, 0] > 0).sum())
    if nck > 3:
        arr, rel = tl[1:nck, :, 0], tl[1:nck, :, 1]
        chunk_len = np.diff(rel.min(axis=1)).mean()
        skew = (arr.max(axis=1) - arr.min(axis=1)).mean()
        last_to_release = (rel.min(axis=1) - arr.max(axis=1)).mean()
        order = np.argsort(arr, axis=1)
        late = np.bincount(order[:, -1], minlength=8)
        # waves 2k, 2k+1?  which waves share a SIMD is the hardware's choice; report arrival offsets per wave
        off = (arr - arr.min(axis=1, keepdims=True)).mean(axis=0)
        print("   block 37: %d chunks, %.0f ticks per chunk; arrival skew (last - first wave) %.0f ticks = %.1f %%, last arrival -> release %.0f ticks; "
              "mean arrival offset per wave %s; last wave counts %s" % (nck, chunk_len, skew, 100.0 * skew / chunk_len, last_to_release,
                                                                         " ".join("%.0f" % o for o in off), late.tolist()))
    print("   shader clock during the loop: %.2f GHz (s_memtime ticks / s_memrealtime at 100 MHz)" % (tot / (s[21] * 10.0)))
    print("%-16s V=%d %.1f us/launch | per wave: loop %.2f us over %.1f chunks = %.0f ns per chunk; waiting for own DMAs %.1f %%, at the barrier %.1f %% of the loop"
          % (name, V, us, tot / waves * 0.01, chunks / waves, tot * 10.0 / max(1, chunks), 100.0 * dma / tot, 100.0 * bar / tot))
