#!/usr/bin/env python3
"""The north star's "within 1e-4 fp32" read against a float64 evaluation of the same algorithm.

For the BASELINE workloads (4 frames x 120 000 points, sigma 0.6; [gru,gru,aflow,gru] and [gru x 4]) every frame's output
of the HIP path is compared with the CPU oracle in fp32 (o32) and in float64 (o64; the pooled PointNet tensor stays the
pinned fp32 fma chain, everything behind it runs in float64) and three max-abs numbers are printed per frame:
|HIP - o64|, |o32 - o64|, |HIP - o32|.  tests/test_gpu_fullsize.py asserts the same numbers; this tool exists for A/B
builds (e.g. TLN_EXTRA_FLAGS=-DTLN_GRU_FAST_GATES, round 3's approximate gates) and writes gpurun_out/parity64_<tag>.json.

  python tools/parity64.py [tag] [points=120000] [direct[=G]]
      direct[=G]: every product on the direct kernel (K split over G waves per tile, partial sums added at the end)
                  instead of gemm_v2's one accumulation chain over all K — the experiment behind DESIGN.md section 2"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests.helpers import build_model, make_config, make_lattice, oracle_pair, randomize_parameters  # noqa: E402
from temporal_latticenet_amd.synthetic import make_sequence  # noqa: E402
from temporal_latticenet_amd import options as OPT  # noqa: E402

OPT.push()   # kernel-selection options of this host thread (tln_options; the library has no process-wide switch)

tag = sys.argv[1] if len(sys.argv) > 1 else "run"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120000
gpu = torch.device("cuda:0")
records = []
for a in sys.argv[3:]:
    if a.startswith("direct"):
        from temporal_latticenet_amd import _lib
        OPT.set(gemm_direct=1)
        if "=" in a:
            OPT.set(gemm_groups=int(a.split("=")[1]))
for rnn in (("gru", "gru", "aflow", "gru"), ("gru", "gru", "gru", "gru")):
    contents = make_config(rnn_modules=rnn, frames=4, sigma=0.6)
    torch.manual_seed(20240607)          # the modules' default initialisation: the same weights on every run
    seq = make_sequence(N, 4)
    model = build_model(contents).eval()
    with torch.no_grad():
        lat = make_lattice(contents)
        for t, (p, v) in enumerate(seq[:2]):
            model(lat, torch.from_numpy(p[:4096]).to(gpu), torch.from_numpy(v[:4096]).to(gpu), t != 1, False)
        model.reset_sequence()
    randomize_parameters(model, seed=5)
    outs = []
    with torch.no_grad():
        lat = make_lattice(contents)
        for t, (p, v) in enumerate(seq):
            a, b, lat = model(lat, torch.from_numpy(p).to(gpu), torch.from_numpy(v).to(gpu), t != 3, False)
            outs.append(b.cpu())
    model.reset_sequence()
    o32, o64 = oracle_pair(model, contents)
    for t, (p, v) in enumerate(seq):
        w32, w64 = o32.forward(p, v, early_return=(t != 3)), o64.forward(p, v, early_return=(t != 3))
        rec = {"case": ",".join(rnn), "frame": t, "points": N, "max_logit": float(w64.abs().max()),
               "hip_vs_o64": float((outs[t].double() - w64).abs().max()),
               "o32_vs_o64": float((w32.double() - w64).abs().max()),
               "hip_vs_o32": float((outs[t] - w32).abs().max()),
               # the typical error beside the worst element
               "hip_vs_o64_rms": float((outs[t].double() - w64).pow(2).mean().sqrt()),
               "o32_vs_o64_rms": float((w32.double() - w64).pow(2).mean().sqrt())}
        records.append(rec)
        print("[parity64 %s] %-16s frame %d  max|x| %6.2f  |HIP-o64| %.3e  |o32-o64| %.3e  |HIP-o32| %.3e   rms %.2e / %.2e"
              % (tag, rec["case"], t, rec["max_logit"], rec["hip_vs_o64"], rec["o32_vs_o64"], rec["hip_vs_o32"],
                 rec["hip_vs_o64_rms"], rec["o32_vs_o64_rms"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "parity64_%s.json" % tag), "w") as f:
    json.dump(records, f, indent=1)
