#!/usr/bin/env python3
"""The single-GPU measurement configurations of SURVEY.md 8(d) (config 4 needs four GPUs: bench.py --mode frames).
Prints one line per configuration: clouds/s with 1 and 4 sequence streams, vertex counts, and for config 1 the CPU
oracle beside it."""
import contextlib, io, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from temporal_latticenet_amd.configs import build_model, make_config, make_lattice
from temporal_latticenet_amd.lattice import Lattice
from temporal_latticenet_amd.streams import SequenceStreams
from temporal_latticenet_amd.synthetic import make_sequence

quiet = lambda: contextlib.redirect_stdout(io.StringIO())


PAIRS_ONLY = len(sys.argv) > 2 and sys.argv[2] == "pairs"


def model_config(name, points, frames, rnn, seq_learning, sigma, capacity, steps):
    contents = make_config(rnn_modules=rnn, sequence_learning=seq_learning, frames=frames, sigma=sigma, capacity=capacity)
    seq = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(points, frames, seed=1234)]
    with quiet(), torch.no_grad():
        torch.manual_seed(1234)
        model = build_model(contents).eval()
        lat = make_lattice(contents)
        for t, (p, v) in enumerate(seq):
            model(lat, p, v, t != frames - 1, False)
        counts = [lat.nr_lattice_vertices()]
        l = lat
        for _ in range(2):
            l = l.coarsen()
            counts.append(l.nr_lattice_vertices())
        model.reset_sequence()
    res = {}
    # the pair mode gets a process of its own (argv[2] == "pairs"): streams created after two earlier pools land on
    # hardware queues that share compute pipes, which costs it 30 %
    runs = ((4, 8),) if PAIRS_ONLY else ((1, False), (4, False))
    for S, pairs in runs:
        with quiet():
            pool = SequenceStreams(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), seq, S,
                                   pairs=pairs)
        per = int(pairs) if pairs else 1          # sequences per stream and step (lock-stepped)
        pool.run([[seq] * (2 * per) for _ in range(S)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pool.run([[seq] * (steps * per) for _ in range(S)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[(S, pairs)] = S * per * steps * frames / dt
        pool.close()
    head = "%-9s %7d pts x %d frames, sigma %.2f, rnn %-22s V0/V1/V2 (last frame) %s :" % (
        name, points, frames, sigma, ",".join(rnn) if seq_learning else "(no sequence learning)", counts)
    if PAIRS_ONLY:
        print(head, "%8.1f clouds/s (4 streams x 8 lock-stepped sequences)" % res[(4, 8)], flush=True)
    else:
        print(head, "%8.1f clouds/s (1 stream) %8.1f clouds/s (4 streams)" % (res[(1, False)], res[(4, False)]), flush=True)


def config1():
    from temporal_latticenet_amd.lattice_modules import ConvLatticeModule, SliceLatticeModule, SplatLatticeModule
    from oracle import ops as O
    from oracle import permuto as P
    n, sigma, C = 20000, 1.0, 32
    pos, val = make_sequence(n, 1, seed=11)[0]
    feat = np.random.default_rng(3).standard_normal((n, C - 1)).astype(np.float32)
    lat = Lattice.from_params([sigma] * 3, 1 << 17)
    splat, conv, slc = SplatLatticeModule(), ConvLatticeModule(C, 1, 1, bias=True), SliceLatticeModule()
    p, f = torch.from_numpy(pos).cuda(), torch.from_numpy(feat).cuda()

    def gpu_once():
        lv, ls, idx, w = splat(lat, p, f)
        lv2, ls = conv(lv, ls)
        return slc(lv2, ls, p, idx, w)

    with torch.no_grad():
        for _ in range(10):
            gpu_once()
        torch.cuda.synchronize()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            gpu_once()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    g = float(np.median(ts))
    W, B = conv.weight.detach().cpu(), conv.bias.detach().cpu()

    def cpu_once():
        tab = P.VertexTable(3, 1 << 17)
        _, oi, ow = O.distribute(tab, pos, feat, [sigma] * 3, subtract_mean=False)
        olv = O.splat(torch.from_numpy(feat), oi, ow, tab.nr_vertices)
        return O.slice_blend(O.conv(olv, P.neighbour_table(tab), W, B), oi, ow)

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cpu_once()
    cs = []
    for _ in range(5):
        t0 = time.perf_counter()
        cpu_once()
        cs.append(time.perf_counter() - t0)
    c = float(np.median(cs))
    print("config 1    20000 pts, sigma 1.00, splat -> conv 9x32->32 -> slice, V = %d : GPU %.3f ms (%.0f clouds/s, latency of one "
          "synchronous cloud), CPU oracle %.1f ms (%.1f clouds/s, %d threads)"
          % (lat.nr_lattice_vertices(), g * 1e3, 1 / g, c * 1e3, 1 / c, torch.get_num_threads()), flush=True)


CONFIGS = {
    "2": ("config 2", 120000, 1, ("gru", "none", "none", "none"), False, 0.6, 1 << 18, 160),
    "3": ("config 3", 120000, 4, ("gru", "gru", "gru", "gru"), True, 0.6, 1 << 18, 80),
    "headline": ("headline", 120000, 4, ("gru", "gru", "aflow", "gru"), True, 0.6, 1 << 18, 80),
    "5": ("config 5", 960000, 1, ("gru", "none", "none", "none"), False, 0.6, 1 << 21, 30),
    "5b": ("config 5b", 120000, 8, ("gru", "gru", "aflow", "gru"), True, 0.6, 1 << 18, 40),
}

if __name__ == "__main__":
    # one configuration per process (stream -> hardware-queue assignment starts fresh):
    #   for c in 1 2 3 headline 5 5b; do python tools/configs_bench.py $c; done
    which = sys.argv[1] if len(sys.argv) > 1 else "1"
    if which == "1":
        config1()
    else:
        model_config(*CONFIGS[which])
