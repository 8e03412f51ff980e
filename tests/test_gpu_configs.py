"""GPU: the measurement configurations of SURVEY.md 8(d) that are not covered elsewhere.
Config 1 — plain splat -> one 9-tap convolution (C = 32) -> plain slice, N = 20 000, sigma = 1.0, no sequence
learning — is the reference's own CPU-runnable case; here the module-level pipeline (SplatLatticeModule,
ConvLatticeModule, SliceLatticeModule) is compared with the oracle on the same weights."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import permuto as P
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


def test_config1_splat_conv_slice_matches_oracle(gpu):
    from temporal_latticenet_amd.lattice import Lattice
    from temporal_latticenet_amd.lattice_modules import ConvLatticeModule, SliceLatticeModule, SplatLatticeModule
    n, sigma, C = 20000, 1.0, 32
    pos, val = make_sequence(n, 1, seed=11)[0]
    feat = np.random.default_rng(3).standard_normal((n, C - 1)).astype(np.float32)     # [values | 1] -> C channels
    lat = Lattice.from_params([sigma] * 3, 1 << 17)
    splat, conv, slc = SplatLatticeModule(), ConvLatticeModule(C, 1, 1, bias=True), SliceLatticeModule()
    with torch.no_grad():
        p, f = torch.from_numpy(pos).to(gpu), torch.from_numpy(feat).to(gpu)
        lv, ls, idx, w = splat(lat, p, f)
        assert lv.shape[1] == C
        lv2, ls = conv(lv, ls)
        out = slc(lv2, ls, p, idx, w)
    # oracle
    tab = P.VertexTable(3, 1 << 17)
    _, oi, ow = O.distribute(tab, pos, feat, [sigma] * 3, subtract_mean=False)
    assert np.array_equal(idx.cpu().numpy(), oi)
    olv = O.splat(torch.from_numpy(feat), oi, ow, tab.nr_vertices)
    table = P.neighbour_table(tab)
    olv2 = O.conv(olv, table, conv.weight.detach().cpu(), conv.bias.detach().cpu())
    want = O.slice_blend(olv2, oi, ow)
    assert out.shape == (n, C)
    scale = max(1.0, float(want.abs().max()))
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=0, atol=1e-4 * scale)
