"""GPU: the HIP fusion modules (same class / parameter names as the reference) vs the golden vectors that the
reference's own classes produced (tests/golden/*.npz)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class TableLattice:
    """`ls` stand-in around an explicit [V,9] neighbour table living on the GPU"""

    def __init__(self, table):
        self.t = torch.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).cuda()
        self._v = None

    def set_values(self, t):
        self._v = t

    def val_dim(self):
        return self._v.shape[1]

    def get_filter_extent(self, n):
        return 9

    def nr_lattice_vertices(self):
        return self.t.shape[0]

    def neighbour_table_ptr(self):
        return C.c_void_p(self.t.data_ptr())


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return z, {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}


def _run(module, z, tol, tables=False, weights=False):
    for t in range(4):
        v = z["x%d" % t].shape[0]
        ls = TableLattice(z["table%d" % t] if tables else np.zeros((v, 9)))
        with torch.no_grad():
            lv, _ = module(torch.from_numpy(z["x%d" % t]).cuda(), ls)
        np.testing.assert_allclose(lv.cpu().numpy(), z["lv%d" % t], rtol=tol, atol=tol)
        if weights and t > 0:
            np.testing.assert_allclose(module.weights_vis.cpu().numpy(), z["w%d" % t], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("c", [64, 128, 192])
def test_gru_module(gpu, c):
    from temporal_latticenet_amd.seq_modules import GRUModule
    z, sd = _load("gru_c%d.npz" % c)
    m = GRUModule(c).cuda()
    m.load_state_dict(sd)
    _run(m, z, 2e-5)


@pytest.mark.parametrize("c", [32, 256])
def test_aflow_module(gpu, c):
    from temporal_latticenet_amd.seq_modules import CrossframeLocalInterpolationModule
    z, sd = _load("aflow_c%d.npz" % c)
    m = CrossframeLocalInterpolationModule(c).cuda()
    # first forward creates AFLOW.weight / AFLOW.bias lazily, like the reference; then the checkpoint is loaded
    x0 = torch.from_numpy(z["x0"]).cuda()
    with torch.no_grad():
        m(x0, TableLattice(z["table0"]))
        m(x0, TableLattice(z["table0"]))
    m.reset_sequence()
    sd_full = dict(sd)
    sd_full["AFLOW.weight"] = m.AFLOW.weight.detach().cpu()
    assert tuple(sd_full["AFLOW.weight"].shape) == tuple(z["shape.AFLOW.weight"])
    m.load_state_dict(sd_full)
    _run(m, z, 1e-4, tables=True, weights=True)


@pytest.mark.parametrize("cls,name,args", [("LSTMModule", "lstm_c64.npz", (64,)),
                                            ("TemporalMaxPoolModule", "maxpool_c64.npz", ()),
                                            ("TemporalLinearModule", "linear_c64.npz", (64,))])
def test_other_fusion_modules(gpu, cls, name, args):
    import temporal_latticenet_amd.seq_modules as S
    z, sd = _load(name)
    m = getattr(S, cls)(*args).cuda()
    m.load_state_dict(sd)
    _run(m, z, 2e-5)


def test_cga_module(gpu):
    """CrossframeGlobalAttentionModule (lm:70-116) on the library's kernels against what the reference's own class
    computed with seeded stand-ins for Conv1x1 / Gn (tests/golden/make_golden.py): hidden_linear, zero padding to V rows,
    conv -> ReLU -> Gn -> the SAME conv, 1 / (V + C), sigmoid, ones for the rows born in this frame, the gate"""
    from temporal_latticenet_amd.seq_modules import CrossframeGlobalAttentionModule
    z, sd = _load("cga_c64.npz")
    m = CrossframeGlobalAttentionModule(64).cuda()
    m.conv._make(64)                                  # the lazily created parameters (created at the first use upstream)
    m.groupnorm.ensure(torch.zeros(1, 64))
    assert set(m.state_dict()) == set(sd), (sorted(m.state_dict()), sorted(sd))
    m.load_state_dict(sd)
    m = m.cuda()
    _run(m, z, 2e-5)
    assert float(np.abs(z["lv3"][53:] - z["x3"][53:]).max()) == 0.0      # rows born in the frame pass through (gate = 1)


def test_pointnet_pool_against_the_reference_vectors(gpu):
    """tests/golden/pointnet_pool.npz came out of the reference's own PointNetSeqModule.forward (lm:407-576, generator
    tests/golden/make_golden.py): -1 indices, an empty vertex, a vertex with < 4 rows, arg-max rows both <= V and > V.
    The HIP pool runs on a lattice that holds the fixture's V vertices (any V distinct keys) and the fixture's indices."""
    from temporal_latticenet_amd import ops
    from temporal_latticenet_amd.lattice import Lattice
    z, sd = _load("pointnet_pool.npz")
    v = int(z["nr_vertices"])
    lat = Lattice.from_params([1.0] * 3, 1 << 12)
    keys = np.zeros((v, 3), np.int32)
    keys[:, 0] = 4 * np.arange(v)                    # lattice points (coordinates congruent mod 4), all different
    lat.insert_keys(torch.from_numpy(keys).cuda())
    assert lat.nr_lattice_vertices() == v
    ws = [sd["layers.%d.weight" % i].cuda() for i in range(3)]
    bs = [sd["layers.%d.bias" % i].cuda() for i in range(3)]
    dist = torch.from_numpy(z["distributed"]).cuda()
    idx = torch.from_numpy(z["indices"].astype(np.int32)).cuda()
    pooled = ops.pointnet_pool(lat, dist, idx, ws, bs, 4).clone()
    pooled[0, :] = 0                                 # lm:569-570 (the fixture's last_conv is an identity stand-in)
    np.testing.assert_allclose(pooled.cpu().numpy(), z["out"], rtol=1e-5, atol=1e-6)
