#!/usr/bin/env python3
"""One sequence at a time, nothing else: the route of `bench.py --streams 1 --pairs 0` without the roofline replay, for a
kernel trace whose every launch belongs to a timed frame.
  python tools/solo_run.py [steps=20]      prints clouds/s; under rocprofv3 --kernel-trace use tools/trace_coverage.py and
  tools/solo_table.py on the trace"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from temporal_latticenet_amd.configs import build_model, make_config, make_lattice  # noqa: E402
from temporal_latticenet_amd.streams import SequenceStreams  # noqa: E402
from temporal_latticenet_amd.workload import group_sequences, stream_drives  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
contents = make_config(capacity=1 << 18)
quiet = contextlib.redirect_stdout(io.StringIO())
with quiet:
    model = build_model(contents).eval()
    frames = None
    drives = stream_drives(120000, 4, 0, 1)
    pool = SequenceStreams(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), drives[0], 1, pairs=False)
per_stream = group_sequences(drives, 1)
with torch.no_grad():
    pool.run([per_stream[0:1] * 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.run([per_stream[0:1] * steps])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("solo: %.1f clouds/s, %.1f us per frame" % (4 * steps / dt, dt / (4 * steps) * 1e6))
