"""Stage-by-stage comparison of the HIP model (operator route) with the oracle on one sequence: prints the max abs
difference after every module, so the first diverging stage is visible.  GPU box only.
  python tools/debug_stages.py [points] [frames] [sigma] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.helpers import build_model, make_config, make_lattice, oracle_from_model, randomize_parameters
from temporal_latticenet_amd.synthetic import make_sequence
from oracle import model as OM, ops as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 31
contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=T, sigma=sigma)
seq = make_sequence(n, T, seed=seed)
torch.manual_seed(20240607)
np.random.seed(20240607)
model = build_model(contents).eval()
model.use_frame_program = False


def run(rec=None):
    lat = make_lattice(contents)
    with torch.no_grad():
        for t, (pos, val) in enumerate(seq):
            if rec is not None:
                rec.append(("frame", t))
            model(lat, torch.from_numpy(pos).cuda(), torch.from_numpy(val).cuda(), t != T - 1, False)
    model.reset_sequence()


run()
randomize_parameters(model, seed=1)
got = []


def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(o):
            got.append((name, o.detach().cpu().clone()))
    return f


for name, mod in model.named_modules():
    if name and name.count(".") <= 2 and not name.endswith(("norm", "conv", "linear", "GRU")):
        mod.register_forward_hook(hook(name))
run(got)

want = []
oracle = oracle_from_model(model, contents)
orig_pool, orig_block, orig_fusion = O.pointnet_pool, OM.OracleLNN._block, OM.OracleLNN._fusion


def pool(*a, **k):
    r = orig_pool(*a, **k)
    want.append(("pool", r.clone()))
    return r


def block(self, lv, table, p):
    r = orig_block(self, lv, table, p)
    want.append((p, r.clone()))
    return r


def fusion(self, slot, kind, lv, table, p):
    r = orig_fusion(self, slot, kind, lv, table, p)
    want.append((p, r.clone()))
    return r


O.pointnet_pool = pool
OM.OracleLNN._block = block
OM.OracleLNN._fusion = fusion
finals = []
for t, (pos, val) in enumerate(seq):
    want.append(("frame", t))
    finals.append(oracle.forward(pos, val, early_return=(t != T - 1)))

gd = {}
fr = None
for name, v in got:
    if name == "frame":
        fr = v
        continue
    gd.setdefault((fr, name), []).append(v)
fr = None
for name, v in want:
    if name == "frame":
        fr = v
        print("== frame", fr)
        continue
    key = (fr, name)
    if name == "pool":
        print("  oracle pool shape", tuple(v.shape))
        continue
    if key in gd and gd[key]:
        g = gd[key].pop(0)
        if g.shape == v.shape:
            d = (g - v).abs()
            r, c = np.unravel_index(int(d.argmax()), d.shape)
            print("  %-48s %s max|d| %.3e at (%d,%d) scale %.2f  rows>1e-3: %d" % (
                name, tuple(v.shape), float(d.max()), r, c, float(v.abs().max()), int((d.max(1).values > 1e-3).sum())))
        else:
            print("  %-48s shape %s vs %s" % (name, tuple(g.shape), tuple(v.shape)))
    else:
        print("  %-48s (no GPU counterpart)" % name)
