"""GPU: BASELINE config 5 scale — an accumulated cloud of 8 x 120k points (kitti_dataloader.py:198-201 concatenates
the frames) hashed at a fine sigma so that the lattice holds ~1M vertices, with the hash capacity raised as
SURVEY.md §8d says (cfg:71's 100000 would overflow).  Indices must stay bit-exact against the sequential oracle at
this size too, and the size-independent properties must hold."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import permuto as P
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


def test_accumulated_cloud_one_million_vertices(gpu):
    from temporal_latticenet_amd import ops
    from temporal_latticenet_amd.lattice import Lattice
    seq = make_sequence(120000, 8, seed=77)
    pos = np.concatenate([p for p, _ in seq], 0)          # accumulate_clouds
    val = np.concatenate([v for _, v in seq], 0)
    sigma, cap = 0.07, 1 << 21
    lat = Lattice.from_params([sigma] * 3, cap)
    d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu))
    V = lat.nr_lattice_vertices()
    assert 900_000 < V < cap and lat.overflow_rows() == 0, V
    tab = P.VertexTable(3, cap)
    od, oi, ow = O.distribute(tab, pos, val, [sigma] * 3)
    assert tab.nr_vertices == V
    assert np.array_equal(i.cpu().numpy(), oi), "bit-exact vertex indices at ~1M vertices"
    assert np.array_equal(w.cpu().numpy(), ow)
    np.testing.assert_allclose(d.cpu().numpy(), od, rtol=0, atol=3e-5)
    # properties that do not need the oracle
    ii = i.cpu().numpy()
    first = np.full(V, ii.shape[0], np.int64)
    np.minimum.at(first, ii, np.arange(ii.shape[0]))
    assert np.all(np.diff(first) > 0), "vertex index order == first-touch order"
    nb = lat.neighbour_table().cpu().numpy()
    v = np.arange(V)
    for tap in range(8):
        u = nb[:, tap]
        ok = u >= 0
        assert np.array_equal(nb[u[ok], tap ^ 1], v[ok])
    # slice(splat(const)) == const
    ones = torch.full((pos.shape[0], 1), 1.5, device=gpu)
    sp = ops.splat(lat, ones, i, w)
    sl = ops.slice_blend(sp, i, w)
    np.testing.assert_allclose((sl[:, 0] / sl[:, 1]).cpu().numpy(), 1.5, rtol=1e-5)
    # a conv on the big level (large-M tile path of the gather-GEMM) against the oracle on a row sample
    g = torch.Generator().manual_seed(0)
    lv = torch.randn(V, 32, generator=g)
    W = torch.randn(9 * 32, 64, generator=g) / np.sqrt(288)
    out = ops.gather_gemm(V, W.to(gpu), ops.gemm_src(lv.to(gpu), lat.neighbour_table_ptr(), 9))
    rows = torch.from_numpy(np.random.default_rng(0).choice(V, 4096, replace=False))
    want = O.im2row(lv, nb[rows.numpy()]) @ W
    np.testing.assert_allclose(out.cpu()[rows].numpy(), want.numpy(), rtol=1e-4, atol=2e-5)


def test_default_capacity_overflow_on_accumulated_cloud_is_reported(gpu):
    from temporal_latticenet_amd.lattice import Lattice
    seq = make_sequence(60000, 4, seed=78)
    pos = np.concatenate([p for p, _ in seq], 0)
    val = np.concatenate([v for _, v in seq], 0)
    lat = Lattice.from_params([0.1] * 3, 100000)           # the reference's default capacity (cfg:71)
    d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu))
    assert lat.nr_lattice_vertices() == 100000
    assert lat.overflow_rows() == int((i < 0).sum()) > 0
