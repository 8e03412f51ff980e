"""GPU parity: lattice structure (K1 distribute, hash numbering, neighbour / cross-level tables) vs oracle.
Integer outputs are compared bit-exactly."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import permuto as P
from temporal_latticenet_amd import options as OPT
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


def _run_sequence(gpu, seq, sigma, capacity, subtract_mean=True, scale_constant=None):
    from temporal_latticenet_amd.lattice import Lattice
    lat = Lattice.from_params([sigma] * 3, capacity, scale_constant=scale_constant)
    tab = P.VertexTable(3, capacity)
    outs = []
    for t, (pos, val) in enumerate(seq):
        d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu),
                                 reset_hashmap=(t == 0), subtract_mean=subtract_mean)
        od, oi, ow = O.distribute(tab, pos, val, [sigma] * 3, subtract_mean, scale_constant)
        outs.append((d.cpu().numpy(), i.cpu().numpy(), w.cpu().numpy(), od, oi, ow))
        assert lat.nr_lattice_vertices() == tab.nr_vertices
        _check_csr(lat, oi)
    return lat, tab, outs


def _check_csr(lat, indices):
    """the handle's CSR == numpy's stable sort of the rows by vertex (rejected rows in the tail bucket V)"""
    V = lat.nr_lattice_vertices()
    key = np.where(indices < 0, V, indices).astype(np.int64)
    want_order = np.argsort(key, kind="stable").astype(np.int32)
    order, sv, seg = (x.cpu().numpy() for x in lat.csr())
    assert np.array_equal(order, want_order), "CSR order must be the stable sort of the rows by vertex"
    assert np.array_equal(sv, key[want_order].astype(np.int32))
    assert np.array_equal(seg, np.searchsorted(key[want_order], np.arange(V + 2)).astype(np.int32))


@pytest.mark.parametrize("rows,nv", [(1, 5), (4097, 3), (300000, 200), (480000, 70000), (1 << 20, 1 << 17)])
def test_build_csr_is_a_stable_sort(gpu, rows, nv):
    """tln_build_csr on caller indices: 1-3 radix passes, ragged last block, -1 rows, heavy duplicates"""
    from temporal_latticenet_amd.lattice import Lattice
    lat = Lattice.from_params([0.5] * 3, 1 << 18)
    rng = np.random.default_rng(rows + nv)
    # a lattice with exactly nv vertices: insert nv distinct keys
    k3 = np.stack([np.arange(nv), -np.arange(nv), np.zeros(nv)], 1).astype(np.int32) * 4
    keys = np.concatenate([k3, -k3.sum(1, keepdims=True)], 1).astype(np.int32)
    lat.insert_keys(torch.from_numpy(keys).to(gpu))
    assert lat.nr_lattice_vertices() == nv
    idx = rng.integers(-1, nv, size=rows).astype(np.int32)
    idx[rng.random(rows) < 0.3] = nv // 2          # one crowded vertex
    for rep in range(2):                           # the arrival counter must be back at zero for the second build
        t = torch.from_numpy(idx).to(gpu)
        lat.ensure_csr(t)
        _check_csr(lat, idx)
        idx = np.roll(idx, 17)


# the lattice scale constant c (scale_i = c / (sigma_i sqrt((i+1)(i+2))), tln_lattice_create_ex) is a free choice of the
# un-vendored dependency (README.md:47): None = Adams' (d+1) sqrt(2/3), the default that meets cfg:71's sizing hint; 1.0 =
# the factor dropped; 2.0 = nobody's choice, the parameter is not a two-way switch
@pytest.mark.parametrize("scale_constant", [None, 1.0, 2.0])
@pytest.mark.parametrize("n,sigma", [(20000, 1.0), (120000, 0.6), (5000, 0.2)])
def test_distribute_matches_oracle(gpu, n, sigma, scale_constant):
    seq = make_sequence(n, 3, seed=7)
    lat, tab, outs = _run_sequence(gpu, seq, sigma, 1 << 18, scale_constant=scale_constant)
    assert abs(lat.scale_constant() - (scale_constant or P.default_scale_constant())) < 1e-12
    lvl = lat
    for _ in range(2):                      # coarse levels inherit the constant
        lvl = lvl.coarsen()
        assert lvl.scale_constant() == lat.scale_constant()
    for d, i, w, od, oi, ow in outs:
        assert np.array_equal(i, oi), "vertex indices must be bit-exact"
        assert np.array_equal(w, ow), "barycentric weights must be bit-exact (same fp32 sequence)"
        np.testing.assert_allclose(d, od, rtol=0, atol=2e-5)
    assert np.array_equal(lat.keys().cpu().numpy(), tab.keys)
    assert lat.overflow_rows() == 0


def test_prefix_stability_and_reset(gpu):
    seq = make_sequence(8000, 2, seed=3)
    lat, tab, outs = _run_sequence(gpu, seq, 0.6, 1 << 16)
    keys_after = lat.keys().cpu().numpy()
    # a fresh run over frame 0 alone gives a prefix of the two-frame numbering
    lat1, tab1, _ = _run_sequence(gpu, seq[:1], 0.6, 1 << 16)
    k1 = lat1.keys().cpu().numpy()
    assert np.array_equal(keys_after[: k1.shape[0]], k1)
    # reset_hashmap really resets
    pos, val = seq[1]
    d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), reset_hashmap=True)
    t2 = P.VertexTable(3, 1 << 16)
    _, oi, _ = O.distribute(t2, pos, val, [0.6] * 3)
    assert np.array_equal(i.cpu().numpy(), oi)


def test_capacity_overflow_is_reported(gpu):
    pos, val = make_sequence(6000, 1, seed=5)[0]
    from temporal_latticenet_amd.lattice import Lattice
    cap = 64
    lat = Lattice.from_params([0.3] * 3, cap)
    d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu))
    tab = P.VertexTable(3, cap)
    _, oi, _ = O.distribute(tab, pos, val, [0.3] * 3)
    i = i.cpu().numpy()
    assert lat.nr_lattice_vertices() == cap
    assert np.array_equal(i, oi)          # first-touch numbering decides which keys fit
    assert lat.overflow_rows() == int((oi < 0).sum()) > 0


def test_neighbour_and_cross_level_tables(gpu):
    seq = make_sequence(30000, 2, seed=11)
    lat, tab, _ = _run_sequence(gpu, seq[:1], 0.6, 1 << 16)
    # level 0 neighbours
    assert np.array_equal(lat.neighbour_table().cpu().numpy(), P.neighbour_table(tab))
    # coarse levels, built incrementally over two frames
    c1 = P.VertexTable(3, 1 << 16)
    c2 = P.VertexTable(3, 1 << 16)
    P.coarsen_insert(c1, tab.keys)
    P.coarsen_insert(c2, c1.keys)
    g1 = lat.coarsen()
    g2 = g1.coarsen()
    assert np.array_equal(g1.keys().cpu().numpy(), c1.keys)
    assert np.array_equal(g2.keys().cpu().numpy(), c2.keys)
    v0, v1 = tab.nr_vertices, c1.nr_vertices
    pos, val = seq[1]
    lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), reset_hashmap=False)
    O.distribute(tab, pos, val, [0.6] * 3)
    P.coarsen_insert(c1, tab.keys[v0:])
    P.coarsen_insert(c2, c1.keys[v1:])
    g1 = lat.coarsen()
    g2 = g1.coarsen()
    assert np.array_equal(g1.keys().cpu().numpy(), c1.keys)
    assert np.array_equal(g2.keys().cpu().numpy(), c2.keys)
    assert np.array_equal(g1.neighbour_table().cpu().numpy(), P.neighbour_table(c1))
    # coarse -> fine taps: fine keys 2*c +- off
    want = tab.lookup(P.neighbour_keys(c1.keys * 2))
    assert np.array_equal(g1.coarse_to_fine_table().cpu().numpy(), want)
    # fine -> coarse taps around the nearest coarse vertex
    want = c1.lookup(P.neighbour_keys(P.finefy_centres(tab.keys)))
    assert np.array_equal(g1.fine_to_coarse_table(tab.nr_vertices).cpu().numpy(), want)


def test_insert_keys_matches_distribute_numbering(gpu):
    pos, val = make_sequence(10000, 1, seed=13)[0]
    from temporal_latticenet_amd.lattice import Lattice
    lat = Lattice.from_params([0.6] * 3, 1 << 16)
    lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu))
    keys = lat.keys()
    lat2 = Lattice.from_params([0.6] * 3, 1 << 16)
    idx = lat2.insert_keys(keys)
    assert np.array_equal(idx.cpu().numpy(), np.arange(keys.shape[0], dtype=np.int32))
    assert np.array_equal(lat2.keys().cpu().numpy(), keys.cpu().numpy())


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 65, 257, 1000])
def test_distribute_fuzz_ragged_sizes_and_scales(gpu, n):
    """clouds of a few points up to a few waves, coordinates from centimetres to hundreds of metres (both signs),
    coarse and fine lattices, duplicated points, two frames (the second one appends): indices, weights and the CSR
    bit-exact, the mean-subtracted rows within float rounding"""
    from temporal_latticenet_amd.lattice import Lattice
    rng = np.random.default_rng(1000 + n)
    for spread, sigma in [(0.05, 0.05), (1.0, 0.5), (30.0, 0.5), (300.0, 3.0), (2.0, 50.0)]:
        lat = Lattice.from_params([sigma] * 3, 1 << 16)
        tab = P.VertexTable(3, 1 << 16)
        for t in range(2):
            pos = (rng.standard_normal((n, 3)) * spread + rng.standard_normal(3) * spread).astype(np.float32)
            if n > 4:
                pos[n // 2] = pos[0]                       # an exact duplicate
            val = rng.random((n, 1)).astype(np.float32)
            d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), reset_hashmap=(t == 0))
            od, oi, ow = O.distribute(tab, pos, val, [sigma] * 3, True)
            assert np.array_equal(i.cpu().numpy(), oi), (n, spread, sigma, t)
            assert np.array_equal(w.cpu().numpy(), ow), (n, spread, sigma, t)
            np.testing.assert_allclose(d.cpu().numpy(), od, rtol=0, atol=1e-5 * max(1.0, spread))
            assert lat.nr_lattice_vertices() == tab.nr_vertices
            _check_csr(lat, oi)
        assert np.array_equal(lat.keys().cpu().numpy(), tab.keys)


@pytest.mark.parametrize("n,spread,sigma", [(1, 1.0, 0.5), (7, 3.0, 0.3), (65, 40.0, 0.5), (900, 200.0, 1.0),
                                            (3000, 5.0, 0.1), (3000, 0.5, 2.0)])
def test_coarse_levels_and_tables_fuzz(gpu, n, spread, sigma):
    """three levels over two frames of random clouds whose keys span both signs: keys, the 9-tap tables of every
    level and both cross-level tables bit-exact"""
    from temporal_latticenet_amd.lattice import Lattice
    rng = np.random.default_rng(7 * n + 1)
    lat = Lattice.from_params([sigma] * 3, 1 << 16)
    tabs = [P.VertexTable(3, 1 << 16) for _ in range(3)]
    seen = [0, 0, 0]
    for t in range(2):
        pos = (rng.standard_normal((n, 3)) * spread - spread * t).astype(np.float32)
        val = rng.random((n, 1)).astype(np.float32)
        lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), reset_hashmap=(t == 0))
        O.distribute(tabs[0], pos, val, [sigma] * 3)
        for l in (1, 2):
            P.coarsen_insert(tabs[l], tabs[l - 1].keys[seen[l - 1]:])
        seen = [tb.nr_vertices for tb in tabs]
        g = [lat, lat.coarsen()]
        g.append(g[1].coarsen())
        for l in range(3):
            assert np.array_equal(g[l].keys().cpu().numpy(), tabs[l].keys), (t, l)
            assert np.array_equal(g[l].neighbour_table().cpu().numpy(), P.neighbour_table(tabs[l])), (t, l)
        for l in (1, 2):
            want = tabs[l - 1].lookup(P.neighbour_keys(tabs[l].keys * 2))
            assert np.array_equal(g[l].coarse_to_fine_table().cpu().numpy(), want), (t, l)
            want = tabs[l].lookup(P.neighbour_keys(P.finefy_centres(tabs[l - 1].keys)))
            assert np.array_equal(g[l].fine_to_coarse_table(tabs[l - 1].nr_vertices).cpu().numpy(), want), (t, l)


def test_scan_ordered_cloud_and_the_bins(gpu):
    """Points in SCAN order (valid / test clouds are not shuffled, kitti_dataloader.py:172): neighbouring rows share their
    vertices, so the wave-level key groups of k_distribute_insert are large (one probe and one row-count atomic per
    group).  Indices and weights bit-exact; the vertex bins hold every row exactly once, grouped by vertex; the pool on
    them equals the oracle bit for bit."""
    from temporal_latticenet_amd import ops
    from temporal_latticenet_amd.lattice import Lattice
    seq = make_sequence(60000, 2, seed=91)
    lat = Lattice.from_params([0.8] * 3, 12000)                       # the second frame overflows the capacity
    tab = P.VertexTable(3, 12000)
    g = torch.Generator().manual_seed(2)
    Ws = [torch.randn(16, 4, generator=g) * 0.5, torch.randn(32, 16, generator=g) * 0.3, torch.randn(64, 32, generator=g) * 0.3]
    Bs = [torch.randn(16, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1]
    for t, (pos, val) in enumerate(seq):
        order = np.lexsort((pos[:, 2], pos[:, 1], np.round(np.arctan2(pos[:, 2], pos[:, 0]), 2)))   # by azimuth, then height
        pos, val = np.ascontiguousarray(pos[order]), np.ascontiguousarray(val[order])
        d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), reset_hashmap=(t == 0))
        od, oi, ow = O.distribute(tab, pos, val, [0.8] * 3)
        assert lat.nr_lattice_vertices() == tab.nr_vertices
        assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(w.cpu().numpy(), ow)
        assert np.array_equal(d.cpu().numpy(), od), "fixed-point means: the distributed rows are bit-exact too"
        assert lat.overflow_rows() == int((oi < 0).sum())
        out = ops.pointnet_pool(lat, d, i, [x.to(gpu) for x in Ws], [x.to(gpu) for x in Bs], 4)
        want = O.pointnet_pool(od, oi, tab.nr_vertices, Ws, Bs, 4)
        assert np.array_equal(out.cpu().numpy(), want.numpy())
    assert (oi < 0).sum() > 0, "the fixture is meant to overflow on the second frame"


@pytest.fixture
def k1_legacy():
    """the per-row-atomic K1 kernels for the duration of a test (the partitioned ones are the default)"""
    from temporal_latticenet_amd import options as OPT
    OPT.push(k1_legacy=1)
    yield OPT
    OPT.pop()


@pytest.mark.parametrize("n,sigma", [(20000, 1.0), (120000, 0.6)])
def test_legacy_k1_matches_oracle(gpu, k1_legacy, n, sigma):
    """tln_options.k1_legacy = 1: k_distribute_insert + the k_bins_* kernels (also what val_dim > 1 takes)"""
    seq = make_sequence(n, 3, seed=7)
    lat, tab, outs = _run_sequence(gpu, seq, sigma, 1 << 18)
    for d, i, w, od, oi, ow in outs:
        assert np.array_equal(i, oi) and np.array_equal(w, ow)
        np.testing.assert_allclose(d, od, rtol=0, atol=2e-5)
    assert np.array_equal(lat.keys().cpu().numpy(), tab.keys)


@pytest.mark.parametrize("n,sigma,capacity,val_dim", [
    (50, 0.6, 1 << 12, 1),          # fewer rows than one bucket workgroup has threads
    (9000, 0.3, 1 << 16, 1),
    (60000, 0.8, 12000, 1),         # the capacity turns keys away (retried, and turned away again, by the next frame)
    (30000, 0.5, 1 << 16, 0),       # positions only
    (300000, 0.05, 1 << 20, 1),     # nearly every row its own vertex: the buckets' LDS tables at their fullest
])
def test_partitioned_and_legacy_k1_agree(gpu, n, sigma, capacity, val_dim):
    """The two K1 variants on the same three frames: indices, weights, rows, keys, overflow count, neighbour table and
    the PointNet pool on their bins are equal bit for bit (the partitioned kernels order the rows inside a vertex's bin
    differently on every run; nothing may depend on it)."""
    from temporal_latticenet_amd import _lib, ops
    from temporal_latticenet_amd.lattice import Lattice
    lib = _lib.lib()
    seq = make_sequence(n, 3, seed=n % 97)
    g = torch.Generator().manual_seed(5)
    Ws = [torch.randn(16, 4, generator=g) * 0.5, torch.randn(32, 16, generator=g) * 0.3, torch.randn(64, 32, generator=g) * 0.3]
    Bs = [torch.randn(16, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1]
    got = {}
    for legacy in (0, 1):
        OPT.push(k1_legacy=legacy)
        try:
            lat = Lattice.from_params([sigma] * 3, capacity)
            res = []
            for t, (pos, val) in enumerate(seq):
                v = torch.from_numpy(val).to(gpu) if val_dim else None
                d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), v, reset_hashmap=(t == 0))
                item = [d.cpu().numpy(), i.cpu().numpy(), w.cpu().numpy(), lat.nr_lattice_vertices(), lat.overflow_rows()]
                if val_dim:
                    item.append(ops.pointnet_pool(lat, d, i, [x.to(gpu) for x in Ws], [x.to(gpu) for x in Bs], 4).cpu().numpy())
                res.append(item)
            got[legacy] = (res, lat.keys().cpu().numpy(), lat.neighbour_table().cpu().numpy())
        finally:
            OPT.pop()
    for a, b in zip(got[0][0], got[1][0]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])
    if capacity == 12000:
        assert got[0][0][-1][4] > 0, "the fixture is meant to overflow"


def test_batched_distribute_of_eight_lattices_equals_the_single_calls(gpu):
    """tln_distribute_begin_multi: the frames of eight lock-stepped sequences (different clouds, different sizes: other
    bucket counts, split-block sizes and threads per split block in one launch) through ONE batch of the four K1
    kernels (blockIdx.y = sequence), two frames each -- indices / weights / keys bit for bit the oracle's, the
    distributed rows and the CSR as from the single calls"""
    from temporal_latticenet_amd.lattice import Lattice
    sizes = [120000, 60000, 5000, 120000, 30000, 1000, 90000, 257]
    seqs = [make_sequence(n, 2, seed=300 + k) for k, n in enumerate(sizes)]
    lats = [Lattice.from_params([0.6] * 3, 1 << 17) for _ in sizes]
    solo = [Lattice.from_params([0.6] * 3, 1 << 17) for _ in sizes]
    tabs = [P.VertexTable(3, 1 << 17) for _ in sizes]
    for t in range(2):
        pos = [torch.from_numpy(s[t][0]).to(gpu) for s in seqs]
        val = [torch.from_numpy(s[t][1]).to(gpu) for s in seqs]
        outs = Lattice.distribute_batch(lats, pos, val, reset_hashmap=(t == 0))
        for k, (d, i, w) in enumerate(outs):
            od, oi, ow = O.distribute(tabs[k], seqs[k][t][0], seqs[k][t][1], [0.6] * 3)
            assert lats[k].nr_lattice_vertices() == tabs[k].nr_vertices, (k, t)
            assert np.array_equal(i.cpu().numpy(), oi), (k, t)
            assert np.array_equal(w.cpu().numpy(), ow), (k, t)
            np.testing.assert_allclose(d.cpu().numpy(), od, rtol=0, atol=2e-5)
            sd, si, sw = solo[k].distribute(pos[k], val[k], reset_hashmap=(t == 0))
            assert torch.equal(d, sd) and torch.equal(i, si) and torch.equal(w, sw), (k, t)
            assert np.array_equal(lats[k].keys().cpu().numpy(), tabs[k].keys)
            _check_csr(lats[k], oi)
            assert lats[k].overflow_rows() == 0


def test_a_bucket_that_overflows_its_table_falls_back_to_the_atomic_kernels(gpu):
    """The partitioned K1 gives every bucket (one workgroup) a 1024-entry LDS table for the distinct keys of a frame.  With
    the buckets made 64x larger than the default (tln_options.k1_bucket_rows) a fine lattice puts thousands of distinct
    keys into each: the kernels notice (their own counter, not the "table too full" one), number nothing, and the library
    redoes the frame with the per-row-atomic kernels -- same indices as the oracle, and the sequence goes on append-only.
    Also through the batched first half, and with the per-row indices switched off for a frame."""
    from temporal_latticenet_amd import _lib
    from temporal_latticenet_amd.lattice import Lattice
    lib = _lib.lib()
    seq = make_sequence(60000, 3, seed=411)
    OPT.push(k1_bucket_rows=32768)
    try:
        lat = Lattice.from_params([0.1] * 3, 1 << 19)
        tab = P.VertexTable(3, 1 << 19)
        for t, (pos, val) in enumerate(seq[:2]):
            d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), reset_hashmap=(t == 0))
            od, oi, ow = O.distribute(tab, pos, val, [0.1] * 3)
            assert lat.bucket_fallbacks() == t + 1, "the frame was meant to overflow a bucket"
            assert lat.nr_lattice_vertices() == tab.nr_vertices > 50000
            assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(w.cpu().numpy(), ow)
            np.testing.assert_allclose(d.cpu().numpy(), od, rtol=0, atol=2e-5)
        OPT.set(k1_bucket_rows=0)                    # the third frame on the default geometry: no fallback, same numbering
        pos, val = seq[2]
        d, i, w = lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), reset_hashmap=False)
        od, oi, ow = O.distribute(tab, pos, val, [0.1] * 3)
        assert lat.bucket_fallbacks() == 2 and np.array_equal(i.cpu().numpy(), oi)
        assert np.array_equal(lat.keys().cpu().numpy(), tab.keys)
        # batched first halves, two lattices, both overflowing
        OPT.set(k1_bucket_rows=32768)
        lats = [Lattice.from_params([0.1] * 3, 1 << 19) for _ in range(2)]
        outs = Lattice.distribute_batch(lats, [torch.from_numpy(seq[k][0]).to(gpu) for k in range(2)],
                                        [torch.from_numpy(seq[k][1]).to(gpu) for k in range(2)])
        for k, (d, i, w) in enumerate(outs):
            t2 = P.VertexTable(3, 1 << 19)
            od, oi, ow = O.distribute(t2, seq[k][0], seq[k][1], [0.1] * 3)
            assert lats[k].bucket_fallbacks() == 1 and np.array_equal(i.cpu().numpy(), oi)
    finally:
        OPT.pop()
