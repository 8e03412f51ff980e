#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own in-tree fusion modules on CPU.

Run in the build container only (needs /root/reference; the GPU box has neither the reference nor a need for this
script — the committed .npz files are the fixtures):

    python tests/golden/make_golden.py

How: /root/reference/seq_lattice/lattice_modules.py imports four third-party modules that are not installed
(`latticenet`, `latticenet_py.lattice.lattice_funcs`, `latticenet_py.lattice.lattice_modules`, `torch_scatter`;
SURVEY.md §8c).  Empty/minimal stand-ins are registered in sys.modules, `.to("cuda")` is made a no-op on this
GPU-less host, and the reference file is imported from where it lies.  All arithmetic AFTER the neighbour gather
(and after scatter_max/scatter_add, whose torch_scatter-2.0.4 semantics are restated in the stand-in) is the
reference's own code; the stand-ins only provide
  * Im2RowLattice / Im2RowIndicesLattice: a gather through an explicit [V,9] neighbour table (centre last, -1 holes)
  * scatter_max / scatter_add: pure-torch segment reductions
  * ConvLatticeModule: identity (so PointNetSeqModule's output is the tensor that enters `last_conv`)
  * for CrossframeGlobalAttentionModule (lm:70-116) only: Conv1x1 = a lazily created, seeded torch.nn.Linear without
    bias applied per vertex (called with ONE argument and returning ONE tensor, as lm:95, 102 call it); Gn = a lazily
    created torch.nn.GroupNorm(32, C) over the lattice in [1, C, V] layout with seeded affine parameters, returning
    (lv, ls) as lm:100 unpacks it.  Parameter names follow the build's modules (conv.linear.weight, groupnorm.norm.*).
    What the fixture pins is the module's own glue: hidden_linear, the zero padding to V rows, conv -> ReLU -> Gn ->
    conv with ONE shared conv weight, the 1 / (V + C) scale, the sigmoid, the ones for rows born in this frame, the gate.
No reference source text is stored in this repository: only inputs, seeded weights and outputs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/seq_lattice/lattice_modules.py"


# ------------------------------------------------------------------------------------------------
# stand-ins
# ------------------------------------------------------------------------------------------------
class FakeLattice:
    def __init__(self, table):
        self.table = torch.as_tensor(table).long()
        self._v = None

    def set_values(self, t):
        self._v = t

    def val_dim(self):
        return self._v.shape[1]

    def get_filter_extent(self, n):
        return 9


def _gather(values, ls):
    t = ls.table
    g = values[t.clamp(min=0)] * (t >= 0)[:, :, None]
    return g.reshape(t.shape[0], -1)


class Im2RowLattice:
    @staticmethod
    def apply(values, ls, filter_extent, dilation, nr_filters):
        return _gather(values, ls)


class Im2RowIndicesLattice:
    @staticmethod
    def apply(values, ls, filter_extent, dilation, nr_filters):
        return ls.table.repeat_interleave(values.shape[1], dim=1)


def scatter_max(src, index, dim=0, dim_size=None):
    rows, c = src.shape
    v = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.full((v, c), float("-inf")).scatter_reduce(0, index[:, None].expand(-1, c), src, "amax", include_self=True)
    rowid = torch.arange(rows)[:, None].expand(-1, c)
    cand = torch.where(src == out[index], rowid, torch.full_like(rowid, rows))
    arg = torch.full((v, c), rows, dtype=torch.long).scatter_reduce(0, index[:, None].expand(-1, c), cand, "amin",
                                                                      include_self=True)
    return torch.where(arg == rows, torch.zeros_like(out), out), arg


def scatter_add(src, index, dim=0, dim_size=None):
    v = int(index.max()) + 1 if dim_size is None else dim_size
    return torch.zeros((v,) + tuple(src.shape[1:])).index_add(0, index, src)


def scatter_mean(src, index, dim=0, dim_size=None, out=None):
    s = scatter_add(src, index, dim, dim_size)
    c = scatter_add(torch.ones(src.shape[0], 1), index, dim, s.shape[0]).clamp(min=1)
    return s / c


class IdentityConv(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, lv, ls):
        return lv, ls


class SeededConv1x1(torch.nn.Module):
    """Conv1x1(out_channels, bias) stand-in for the CGA fixture: per-vertex Linear, created at the first call"""

    def __init__(self, out_channels=None, bias=True, *a, **k):
        super().__init__()
        self.out_channels, self.use_bias, self.linear = out_channels, bias, None

    def forward(self, lv, ls=None):
        if self.linear is None:
            self.linear = torch.nn.Linear(lv.shape[1], self.out_channels, bias=self.use_bias)
        out = self.linear(lv)
        return out if ls is None else (out, ls)


class SeededGn(torch.nn.Module):
    """Gn() stand-in for the CGA fixture: GroupNorm(32 groups, or C/2) over all vertices of a group ([1, C, V] layout)"""

    def __init__(self, *a, **k):
        super().__init__()
        self.norm = None

    def forward(self, lv, ls):
        if self.norm is None:
            c = lv.shape[1]
            self.norm = torch.nn.GroupNorm(32 if c % 32 == 0 else c // 2, c)
            with torch.no_grad():
                self.norm.weight.copy_(torch.rand(c) * 0.5 + 0.75)
                self.norm.bias.copy_(torch.randn(c) * 0.05)
        return self.norm(lv.t().unsqueeze(0)).squeeze(0).t(), ls


def install_stand_ins():
    m = types.ModuleType("latticenet")
    m.HashTable = type("HashTable", (), {})
    m.Lattice = type("Lattice", (), {})
    sys.modules["latticenet"] = m
    ts = types.ModuleType("torch_scatter")
    ts.scatter_max, ts.scatter_add, ts.scatter_mean = scatter_max, scatter_add, scatter_mean
    sys.modules["torch_scatter"] = ts
    pkg = types.ModuleType("latticenet_py")
    sub = types.ModuleType("latticenet_py.lattice")
    funcs = types.ModuleType("latticenet_py.lattice.lattice_funcs")
    funcs.Im2RowLattice, funcs.Im2RowIndicesLattice = Im2RowLattice, Im2RowIndicesLattice
    mods = types.ModuleType("latticenet_py.lattice.lattice_modules")
    mods.ConvLatticeModule = IdentityConv
    mods.Gn = IdentityConv
    mods.Conv1x1 = IdentityConv
    mods.GnRelu1x1 = IdentityConv
    for name, mod in [("latticenet_py", pkg), ("latticenet_py.lattice", sub),
                      ("latticenet_py.lattice.lattice_funcs", funcs), ("latticenet_py.lattice.lattice_modules", mods)]:
        sys.modules[name] = mod
    # "cuda" -> no-op on this GPU-less host
    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
        k = {kk: vv for kk, vv in k.items() if not (kk == "device" and str(vv).startswith("cuda"))}
        return orig_to(self, *a, **k) if (a or k) else self

    torch.Tensor.to = to
    orig_mod_to = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_mod_to(self, *a, **k)
    torch.cuda.FloatTensor = torch.FloatTensor


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_lattice_modules", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------
# data
# ------------------------------------------------------------------------------------------------
V_T = [37, 53, 53, 80]           # growing lattice, including a "no growth" step


def neighbour_table(v, rng):
    """random [v,9] table: 8 neighbours with ~35% holes (-1), centre LAST = the vertex itself"""
    t = rng.integers(0, v, size=(v, 9))
    holes = rng.uniform(size=(v, 9)) < 0.35
    t[holes] = -1
    t[:, 8] = np.arange(v)
    return t.astype(np.int64)


def sd_np(module):
    out = {}
    for k, v in module.state_dict().items():
        if k.endswith("AFLOW.weight"):
            # created and initialised by the reference but never used (lm:291-295): keep its SHAPE only
            out["shape." + k] = np.array(v.shape, np.int64)
        else:
            out["sd." + k] = v.detach().numpy().copy()
    return out


def run_fusion(ref, make, c, seed, with_tables=False):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    m = make()
    out = {}
    tables = []
    xs = [torch.randn(v, c) for v in V_T]
    for t, x in enumerate(xs):
        tab = neighbour_table(x.shape[0], rng)
        if t > 0 and with_tables:
            # a vertex that existed at t-1 keeps its neighbours; only rows of new vertices are fresh
            prev = tables[-1]
            tab[: prev.shape[0]] = np.where(prev >= 0, prev, tab[: prev.shape[0]])
            tab[:, 8] = np.arange(x.shape[0])
        tables.append(tab)
        ls = FakeLattice(tab)
        with torch.no_grad():
            lv, _ = m(x.clone(), ls)
        out["x%d" % t] = x.numpy()
        out["lv%d" % t] = lv.numpy().copy()
        if with_tables:
            out["table%d" % t] = tab
            if t > 0:
                out["w%d" % t] = m.weights_vis.numpy().copy()
    out.update(sd_np(m))        # after the run: lazily created parameters exist now
    return out


def main():
    install_stand_ins()
    ref = load_reference()
    os.makedirs(HERE, exist_ok=True)
    for c in (64, 128, 192):
        np.savez_compressed(os.path.join(HERE, "gru_c%d.npz" % c), **run_fusion(ref, lambda: ref.GRUModule(c), c, 100 + c))
    np.savez_compressed(os.path.join(HERE, "lstm_c64.npz"), **run_fusion(ref, lambda: ref.LSTMModule(64), 64, 7))
    np.savez_compressed(os.path.join(HERE, "maxpool_c64.npz"), **run_fusion(ref, lambda: ref.TemporalMaxPoolModule(), 64, 8))
    np.savez_compressed(os.path.join(HERE, "linear_c64.npz"), **run_fusion(ref, lambda: ref.TemporalLinearModule(64), 64, 9))
    for c in (32, 256):
        np.savez_compressed(os.path.join(HERE, "aflow_c%d.npz" % c),
                            **run_fusion(ref, lambda: ref.CrossframeLocalInterpolationModule(c), c, 300 + c, True))
    # CrossframeGlobalAttentionModule (lm:70-116) with seeded stand-ins for its two un-vendored sub-modules
    ref.Conv1x1, ref.Gn = SeededConv1x1, SeededGn
    np.savez_compressed(os.path.join(HERE, "cga_c64.npz"),
                        **run_fusion(ref, lambda: ref.CrossframeGlobalAttentionModule(64), 64, 21))
    ref.Conv1x1 = ref.Gn = IdentityConv

    # PointNetSeqModule: distributed [4N,5], indices with -1s, a vertex with < 4 rows, winning rows both <= V and > V
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    n, v = 300, 41
    dist = torch.randn(4 * n, 5)
    dist[:, 4] = torch.rand(4 * n)
    idx = rng.integers(0, v, size=4 * n)
    idx[rng.uniform(size=4 * n) < 0.03] = -1
    idx[idx == 5] = 6                      # vertex 5 gets no rows at all
    idx[np.nonzero(idx == 7)[0][2:]] = 8   # vertex 7 keeps only 2 rows (< 4 -> masked)
    idx[:6] = [40, 40, 40, 40, 40, 3]      # the last vertex exists; early rows (ids <= V) can win
    dist[:5, :4] += 50.0                   # ... and do win for vertex 40 => argmax <= V branch of lm:514
    m = ref.PointNetSeqModule([16, 32, 64], 64, "none", ["gru", "none", "none", "none"], sequence_learning=False)
    ls = FakeLattice(neighbour_table(v, rng))
    with torch.no_grad():
        out, _ = m(ls, dist.clone(), torch.from_numpy(idx.astype(np.int32)))
    d = {"distributed": dist.numpy(), "indices": idx.astype(np.int32), "nr_vertices": np.int64(v), "out": out.numpy()}
    d.update({k: val for k, val in sd_np(m).items() if "layers" in k})
    np.savez_compressed(os.path.join(HERE, "pointnet_pool.npz"), **d)
    print("golden vectors written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("  %-22s %7d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
