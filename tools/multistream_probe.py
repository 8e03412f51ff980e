#!/usr/bin/env python3
"""Aggregate throughput of S independent sequence streams on one GPU (one host thread + HIP stream + model replica
each).  The per-frame work is latency-bound (a few thousand vertices), so concurrent streams fill the idle CUs."""
import contextlib, io, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from temporal_latticenet_amd.configs import build_model, make_config, make_lattice
from temporal_latticenet_amd.synthetic import make_sequence

S_LIST = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4").split(",")]
STEPS = 20
contents = make_config(capacity=1 << 18)
frames = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(120000, 4)]


def run_sequence(model, lat, fr):
    for t, (p, v) in enumerate(fr):
        model(lat, p, v, t != 3, False)
    model.reset_sequence()


with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(1234)
    base = build_model(contents).eval()
    with torch.no_grad():
        run_sequence(base, make_lattice(contents), frames)
sd = base.state_dict()

for S in S_LIST:
    models, lats, streams = [], [], []
    for s in range(S):
        with contextlib.redirect_stdout(io.StringIO()):
            m = build_model(contents).eval()
            with torch.no_grad():
                run_sequence(m, make_lattice(contents), frames)
            m.load_state_dict(sd)
        models.append(m)
        lats.append(make_lattice(contents))
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    barrier = threading.Barrier(S + 1)

    def worker(i):
        with torch.no_grad(), torch.cuda.stream(streams[i]):
            for _ in range(3):
                run_sequence(models[i], lats[i], frames)
            streams[i].synchronize()
            barrier.wait()
            for _ in range(STEPS):
                run_sequence(models[i], lats[i], frames)
            streams[i].synchronize()
            barrier.wait()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    print("streams %d: %.1f clouds/s  (%.3f ms per sequence per stream)" % (S, S * STEPS * 4 / dt, dt / STEPS * 1e3), flush=True)
