#!/usr/bin/env python3
"""Which batched stage of group mode changes a bit against the solo run?  8 lock-stepped 120k-point sequences through
forward_group with one op kind at a time taken out of the batching (tln_program_group_config), compared with the solo
runs held on the group's kernels (tln_gemm_v2_config(0, 1))."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from temporal_latticenet_amd import _lib  # noqa: E402
from temporal_latticenet_amd.configs import build_model, make_config, make_lattice  # noqa: E402
from temporal_latticenet_amd.models import forward_group  # noqa: E402
from temporal_latticenet_amd.streams import share_parameters  # noqa: E402
from temporal_latticenet_amd.synthetic import make_sequence  # noqa: E402
from temporal_latticenet_amd.workload import turned  # noqa: E402
from tests.helpers import randomize_parameters  # noqa: E402
from temporal_latticenet_amd import options as OPT  # noqa: E402

OPT.push()   # kernel-selection options of this host thread (tln_options; the library has no process-wide switch)

KINDS = {0: "K1/levels/tables", 2: "GN_PARTIALS", 3: "POOL", 4: "GRU", 5: "AFLOW", 8: "COPY", 11: "SLICE_DEFORM"}


def main():
    G, T, N = 8, int(os.environ.get("T", "2")), 120000
    contents = make_config(frames=T, sigma=0.6, capacity=1 << 18)
    drive = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(N, T)]
    seqs = [turned(drive, j) for j in range(G)]
    lib = _lib.lib()
    models = []
    with torch.no_grad():
        for k in range(G):
            m = build_model(contents).eval()
            lat = make_lattice(contents)
            for t, (p, v) in enumerate(drive):
                m(lat, p[:4096], v[:4096], t != T - 1, False)
            m.reset_sequence()
            if k == 0:
                randomize_parameters(m, 3)
            else:
                share_parameters(m, models[0])
            models.append(m)

        def solo(k):
            lat = make_lattice(contents)
            outs = []
            for t, (p, v) in enumerate(seqs[k]):
                a, b, lat = models[0](lat, p, v, t != T - 1, False)
                outs.append(b.clone())
            models[0].reset_sequence()
            return outs

        OPT.set(v2_off=0, v2_min_m=1)
        want = [solo(k) for k in range(G)]
        OPT.set(v2_off=0, v2_min_m=0)

        def group():
            lats = [make_lattice(contents) for _ in range(G)]
            outs = [[] for _ in range(G)]
            for t in range(T):
                res = forward_group(models, lats, [s[t][0] for s in seqs], [s[t][1] for s in seqs], t != T - 1)
                lats = [r[2] for r in res]
                for k in range(G):
                    outs[k].append(res[k][1].clone())
            for m in models:
                m.reset_sequence()
            return outs

        for mask_name, mask in [("all batched", 0)] + [("without " + nm, 1 << k) for k, nm in KINDS.items()] + \
                [("nothing batched", sum(1 << k for k in KINDS))]:
            OPT.set(group_off_mask=mask)
            got = group()
            worst = [max(float((got[k][t] - want[k][t]).abs().max()) for k in range(G)) for t in range(T)]
            print("%-28s max |group - solo| per frame: %s" % (mask_name, " ".join("%.3e" % w for w in worst)), flush=True)
        OPT.set(group_off_mask=0)


if __name__ == "__main__":
    main()
