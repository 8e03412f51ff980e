#!/usr/bin/env python3
"""cProfile of the Python host path of one bench step (where does the host time go?)"""
import contextlib, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from temporal_latticenet_amd.configs import build_model, make_config, make_lattice
from temporal_latticenet_amd.synthetic import make_sequence

contents = make_config(capacity=1 << 18)
with contextlib.redirect_stdout(io.StringIO()):
    model = build_model(contents).eval()
lat = make_lattice(contents)
frames = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(120000, 4)]

def step():
    l = lat
    for t, (p, v) in enumerate(frames):
        a, b, l = model(l, p, v, t != 3, False)
    model.reset_sequence()

with torch.no_grad():
    with contextlib.redirect_stdout(io.StringIO()):
        step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.3f ms/step, +drain %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t1) * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:50]))
