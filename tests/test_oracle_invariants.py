"""CPU: known-answer tests of the lattice oracle from the mathematics of the permutohedral lattice
(SURVEY.md §8c "known-answer tests available without any reference")."""
import numpy as np
import torch

from oracle import ops as O
from oracle import permuto as P
from temporal_latticenet_amd.synthetic import make_sequence


import pytest


def _keys(pos, sigma, constant=None):
    rem0, rank, bary = P.simplex(P.elevate(pos, P.scale_factors([sigma] * 3, constant)))
    return P.simplex_keys(rem0, rank), bary, rem0, rank


@pytest.mark.parametrize("constant", [None, 1.0, 2.0])
def test_barycentric_weights_are_a_partition_of_unity(constant):
    pos = make_sequence(20000, 1, seed=1)[0][0]
    keys, bary, _, _ = _keys(pos, 0.6, constant)
    assert bary[:, :4].min() >= -1e-5
    np.testing.assert_allclose(bary[:, :4].sum(1), 1.0, atol=1e-5)


def test_keys_are_lattice_points():
    pos = (np.random.default_rng(0).normal(size=(5000, 3)) * 30).astype(np.float32)
    keys, _, rem0, rank = _keys(pos, 0.45)
    full = np.concatenate([keys, -keys.sum(-1, keepdims=True)], -1)       # implied last coordinate
    r = np.arange(4)[None, :, None]
    assert np.all((full - r) % 4 == 0), "all coordinates of vertex r are congruent to r mod d+1"
    assert sorted(set(rank.reshape(-1).tolist())) == [0, 1, 2, 3]
    assert np.all(np.sort(rank, axis=1) == np.arange(4)[None, :]), "ranks are a permutation"


def test_consecutive_simplex_vertices_are_one_hop_neighbours():
    """vertices r and r+1 (mod d+1) of a simplex differ by one axis offset (1,..,1,-d,1,..); r and r+2 differ by
    (2,2,-2,-2) and are NOT in the 2(d+1) one-hop neighbourhood"""
    pos = make_sequence(3000, 1, seed=2)[0][0]
    keys, _, _, _ = _keys(pos, 0.8)
    tab = P.VertexTable(3, 1 << 16)
    idx = tab.insert(keys.reshape(-1, 3)).reshape(-1, 4)
    nbr = P.neighbour_table(tab)
    for a in range(4):
        b = (a + 1) % 4
        assert np.all((nbr[idx[:, a], :8] == idx[:, b][:, None]).any(1))
        assert np.all((nbr[idx[:, b], :8] == idx[:, a][:, None]).any(1))
        c = (a + 2) % 4
        assert not np.any((nbr[idx[:, a], :8] == idx[:, c][:, None]).any(1))


def test_neighbour_relation_is_symmetric_with_paired_taps():
    pos = make_sequence(8000, 1, seed=3)[0][0]
    keys, _, _, _ = _keys(pos, 0.6)
    tab = P.VertexTable(3, 1 << 16)
    tab.insert(keys.reshape(-1, 3))
    nbr = P.neighbour_table(tab)
    v = np.arange(nbr.shape[0])
    assert np.array_equal(nbr[:, 8], v), "centre is the last tap"
    for tap in range(8):
        u = nbr[:, tap]
        ok = u >= 0
        assert np.array_equal(nbr[u[ok], tap ^ 1], v[ok]), "tap 2a and 2a+1 are inverse steps"


def test_slice_of_splat_reproduces_a_constant_field():
    pos, val = make_sequence(5000, 1, seed=4)[0]
    tab = P.VertexTable(3, 1 << 16)
    d, idx, w = O.distribute(tab, pos, val, [0.7] * 3)
    ones = torch.full((pos.shape[0], 1), 2.5)
    lv = O.splat(ones, idx, w, tab.nr_vertices)
    sl = O.slice_blend(lv, idx, w)
    np.testing.assert_allclose((sl[:, 0] / sl[:, 1]).numpy(), 2.5, rtol=1e-5)


def test_numbering_is_prefix_stable_and_first_touch():
    seq = make_sequence(4000, 2, seed=5)
    tab = P.VertexTable(3, 1 << 16)
    d0, i0, _ = O.distribute(tab, seq[0][0], seq[0][1], [0.6] * 3)
    k0 = tab.keys.copy()
    # first touch: a vertex's index is the number of distinct keys seen before its first row
    first_seen = {}
    for r, i in enumerate(i0.tolist()):
        first_seen.setdefault(i, r)
    order = sorted(first_seen, key=first_seen.get)
    assert order == list(range(len(order)))
    O.distribute(tab, seq[1][0], seq[1][1], [0.6] * 3)
    assert np.array_equal(tab.keys[: k0.shape[0]], k0)


def test_doubling_sigma_halves_the_elevated_coordinates():
    pos = make_sequence(1000, 1, seed=6)[0][0]
    e1 = P.elevate(pos, P.scale_factors([0.5] * 3))
    e2 = P.elevate(pos, P.scale_factors([1.0] * 3))
    np.testing.assert_allclose(e1, 2 * e2, rtol=1e-5, atol=1e-5)


def test_coarse_embedding_is_integer_exact_and_consistent():
    pos = make_sequence(6000, 1, seed=7)[0][0]
    keys, _, _, _ = _keys(pos, 0.6)
    tab = P.VertexTable(3, 1 << 16)
    tab.insert(keys.reshape(-1, 3))
    ck, bn = P.coarse_simplex_int(tab.keys)
    assert np.all(bn >= 0) and np.all(bn.sum(1) == 8), "barycentric numerators sum to 2(d+1)"
    # the integer construction equals the float simplex search on f/2 wherever the float one has no tie
    f = np.concatenate([tab.keys, -tab.keys.sum(1, keepdims=True)], 1).astype(np.float32) * 0.5
    rem0, rank, bary = P.simplex(f)
    np.testing.assert_allclose(bary[:, :4], bn / 8.0, atol=1e-6)
    # a fine vertex with all-even coordinates that is itself a coarse lattice point embeds with weight 1
    full = np.concatenate([tab.keys, -tab.keys.sum(1, keepdims=True)], 1)
    on_coarse = np.all(full % 8 == (full[:, :1] % 8), axis=1) & np.all(full % 2 == 0, axis=1)
    assert on_coarse.any()
    assert np.all(bn[on_coarse].max(1) == 8)
    # finefy centre of such a vertex is the vertex itself at half resolution
    c = P.finefy_centres(tab.keys[on_coarse])
    assert np.array_equal(c * 2, tab.keys[on_coarse])
