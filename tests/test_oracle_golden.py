"""CPU: the oracle's restatement of the reference's in-tree modules vs the golden vectors produced by the
reference's own classes (tests/golden/make_golden.py).  This is what pins the oracle for the fusion path."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = 4


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    return z, sd


@pytest.mark.parametrize("c", [64, 128, 192])
def test_gru_matches_reference(c):
    z, sd = _load("gru_c%d.npz" % c)
    h = None
    for t in range(FRAMES):
        lv, h = O.gru_step(torch.from_numpy(z["x%d" % t]), h, sd)
        np.testing.assert_allclose(lv.numpy(), z["lv%d" % t], rtol=1e-5, atol=1e-6)
    assert set(sd) == {"GRU.weight_ih", "GRU.weight_hh", "GRU.bias_ih", "GRU.bias_hh", "hidden_linear.weight",
                       "hidden_linear.bias"}
    assert tuple(sd["GRU.weight_ih"].shape) == (3 * c, c)


@pytest.mark.parametrize("c", [32, 256])
def test_aflow_matches_reference(c):
    z, sd = _load("aflow_c%d.npz" % c)
    h = None
    for t in range(FRAMES):
        lv, h, w = O.aflow_step(torch.from_numpy(z["x%d" % t]), h, z["table%d" % t], sd)
        np.testing.assert_allclose(lv.numpy(), z["lv%d" % t], rtol=2e-5, atol=2e-5)
        if t > 0:
            np.testing.assert_allclose(w.numpy(), z["w%d" % t], rtol=1e-5, atol=1e-7)
    assert tuple(z["shape.AFLOW.weight"]) == (9 * c, c)      # registered, never used (lm:291)
    assert {"AFLOW.alpha", "AFLOW.beta", "AFLOW.bias", "linear.weight", "linear.bias"} <= set(sd)


def _run_model_fusion(kind, z, sd):
    from oracle.model import OracleLNN
    m = OracleLNN({("f." + k): v for k, v in sd.items()}, 2, ["none"] * 4)
    outs = []
    for t in range(FRAMES):
        outs.append(m._fusion("slot", kind, torch.from_numpy(z["x%d" % t]), None, "f"))
    return outs


@pytest.mark.parametrize("kind,name", [("lstm", "lstm_c64.npz"), ("maxpool", "maxpool_c64.npz"),
                                       ("linear", "linear_c64.npz"), ("cga", "cga_c64.npz")])
def test_other_fusion_modules_match_reference(kind, name):
    """cga: lm:70-116 run by the reference's own class with seeded stand-ins for its two un-vendored sub-modules
    (tests/golden/make_golden.py: Conv1x1 = Linear without bias, Gn = GroupNorm(32, C) over [1, C, V])"""
    z, sd = _load(name)
    for t, lv in enumerate(_run_model_fusion(kind, z, sd)):
        np.testing.assert_allclose(lv.numpy(), z["lv%d" % t], rtol=1e-5, atol=1e-6)


def test_pointnet_pool_matches_reference():
    z, sd = _load("pointnet_pool.npz")
    ws = [sd["layers.%d.weight" % i] for i in range(3)]
    bs = [sd["layers.%d.bias" % i] for i in range(3)]
    v = int(z["nr_vertices"])
    pooled = O.pointnet_pool(z["distributed"], z["indices"], v, ws, bs, 4)
    # the reference module goes on: no fusion (sequence_learning False), row 0 zeroed (lm:569-570), then last_conv
    # (identity stand-in in the generator)
    pooled = pooled.clone()
    pooled[0, :] = 0
    np.testing.assert_allclose(pooled.numpy(), z["out"], rtol=1e-5, atol=1e-6)
    # the fixture really exercises the quirks: -1 indices, an empty vertex, a masked (<4 rows) vertex
    idx = z["indices"]
    assert (idx < 0).any() and not (idx == 5).any() and (idx == 7).sum() == 2
    assert np.all(z["out"][5] == 0) and np.all(z["out"][7] == 0)
