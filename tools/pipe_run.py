#!/usr/bin/env python3
"""One sequence at a time with its frames pipelined over four slots of ONE GPU (pipeline.FramePipeline) against the
sequential route: clouds/s and the latency of a sequence.   python tools/pipe_run.py [steps=20]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

from temporal_latticenet_amd.configs import build_model, make_config, make_lattice  # noqa: E402
from temporal_latticenet_amd.pipeline import FramePipeline  # noqa: E402
from temporal_latticenet_amd.workload import stream_drives  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
contents = make_config(capacity=1 << 18)
quiet = contextlib.redirect_stdout(io.StringIO())
seq = stream_drives(120000, 4, 0, 1)[0]
with quiet:
    torch.manual_seed(1234)
    model = build_model(contents).eval()


lat0 = make_lattice(contents)


def sequential(n):
    for _ in range(n):
        lat = lat0                                  # (one lattice, cleared by the first frame of every sequence)
        for t, (p, v) in enumerate(seq):
            a, b, lat = model(lat, p, v, t != 3, False)
        model.reset_sequence()
    return b


with torch.no_grad():
    sequential(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    want = sequential(steps).clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("sequential: %.1f clouds/s, %.2f ms per sequence" % (4 * steps / dt, dt / steps * 1e3))
    with quiet:
        pipe = FramePipeline(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), seq)
    for _ in range(3):
        pipe.run([seq], keep_outputs=False)
    torch.cuda.synchronize()
    lat_ms = []
    for _ in range(steps):                       # one sequence in flight: latency
        t0 = time.perf_counter()
        out = pipe.run([seq])
        torch.cuda.synchronize()
        lat_ms.append((time.perf_counter() - t0) * 1e3)
    same = torch.equal(out[0][1], want)
    lat_ms.sort()
    print("pipelined, one sequence in flight: %.2f ms per sequence (median; best %.2f) = %.1f clouds/s; bitwise = sequential: %s"
          % (lat_ms[len(lat_ms) // 2], lat_ms[0], 4e3 / lat_ms[len(lat_ms) // 2], same))
    # the wavefront of one sequence: marks on every slot's stream, in ms after the start of the sequence
    for sl in pipe.slots:
        sl.trace = []
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    pipe.run([seq])
    torch.cuda.synchronize()
    for g, sl in enumerate(pipe.slots):
        print("  slot %d: %s" % (g, "  ".join("%s %.2f" % (lab, e0.elapsed_time(ev)) for lab, ev in sl.trace)))
        sl.trace = None
    t0 = time.perf_counter()
    pipe.run([seq] * steps, keep_outputs=False)   # sequences back to back through the slots
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pipelined, sequences back to back: %.1f clouds/s" % (4 * steps / dt))
    pipe.close()
