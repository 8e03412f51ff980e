#!/usr/bin/env python3
"""One stream stepping a lock-step group of G sequences, nothing else: every K1 / K2 / K8 launch of the run carries the frames
of G sequences (blockIdx.y = sequence), as in bench.py's timed mode — for per-kernel time and HBM-side bytes of the BATCHED
stages (bash tools/kprof.sh <out> 'k_bk|k_pool|k_slice' tools/group_run.py 6; bytes per frame = per launch / G).
  python tools/group_run.py [steps=6] [G=8]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from temporal_latticenet_amd.configs import build_model, make_config, make_lattice, suggest_capacity  # noqa: E402
from temporal_latticenet_amd.streams import SequenceStreams  # noqa: E402
from temporal_latticenet_amd.workload import group_sequences, stream_drives  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
contents = make_config(capacity=suggest_capacity(120000, 0.6, 4))
with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(1234)
    model = build_model(contents).eval()
    drives = stream_drives(120000, 4, 1234, 1)
    with torch.no_grad():               # the lazily created parameters of the base model
        lat = make_lattice(contents)
        for t, (p, v) in enumerate(drives[0][:2]):
            model(lat, p[:4096], v[:4096], t != 1, False)
        model.reset_sequence()
    # (the pool's constructor warms its model replicas on 4096-point slices of the first drive: their K1 launches are
    #  single-frame and tiny; everything the timed steps launch is batched)
    pool = SequenceStreams(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), drives[0], 1, pairs=G)
seqs = group_sequences(drives, G)
with torch.no_grad():
    pool.run([seqs[:G]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.run([seqs[:G] * steps])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("group of %d: %.1f clouds/s (%d steps, %d frames per batched launch)" % (G, 4 * G * steps / dt, steps, G))
pool.close()
