"""GPU: independent sequences on concurrent HIP streams (temporal_latticenet_amd/streams.py) compute exactly what each
would compute alone — same kernels, per-stream lattices, hidden states and workspaces."""
import pytest
import torch

from tests.helpers import build_model, make_config, make_lattice, randomize_parameters
from temporal_latticenet_amd import options as O
from temporal_latticenet_amd.streams import SequenceStreams
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


def _alone(model, contents, seq):
    lat = make_lattice(contents)
    with torch.no_grad():
        for t, (p, v) in enumerate(seq):
            a, b, lat = model(lat, p, v, t != len(seq) - 1, False)
    model.reset_sequence()
    return b.clone()


@pytest.mark.parametrize("frame_program", [True, False])
def test_concurrent_streams_equal_the_single_stream_results(gpu, frame_program):
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=3, sigma=0.7)
    S = 3
    seqs = [[(torch.from_numpy(p).to(gpu), torch.from_numpy(v).to(gpu)) for p, v in make_sequence(9000 + 500 * s, 3, seed=70 + s)]
            for s in range(2 * S)]
    model = build_model(contents).eval()
    _alone(model, contents, seqs[0])
    randomize_parameters(model, seed=9)
    model.use_frame_program = frame_program
    pool = SequenceStreams(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), seqs[0], S)
    want = [_alone(model, contents, s) for s in seqs]       # after the pool: same process-wide kernel choices
    for m in pool.models:
        m.use_frame_program = frame_program
    assert pool.models[1].point_net_seq.layers[0].weight is model.point_net_seq.layers[0].weight
    for rep in range(3):
        got = pool.run([[seqs[i], seqs[S + i]] for i in range(S)], keep_outputs=True)
        for i in range(S):
            assert torch.equal(got[i][0], want[i]), "stream %d, first sequence" % i
            assert torch.equal(got[i][1], want[S + i]), "stream %d, second sequence" % i
    if frame_program:
        assert all(getattr(m, "_program", None) is not None for m in pool.models)
    pool.close()


def test_lockstep_pairs_match_the_single_sequence_results(gpu):
    """two sequences per stream in lock-step (shared gather-GEMM launches): every sequence's scores equal its solo
    run up to the K-summation order of the paired products; with the pairing switched off (two launches per product)
    they are bitwise equal, which pins the lock-step host logic by itself"""
    from temporal_latticenet_amd import _lib
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=3, sigma=0.7)
    S = 2
    seqs = [[(torch.from_numpy(p).to(gpu), torch.from_numpy(v).to(gpu)) for p, v in make_sequence(9000 + 700 * s, 3, seed=170 + s)]
            for s in range(2 * S + 1)]
    model = build_model(contents).eval()
    _alone(model, contents, seqs[0])
    randomize_parameters(model, seed=19)
    # (the bitwise half needs the same kernel on both routes: products sharing a launch may take the large-M kernel
    # where each alone takes the direct one — their rows are counted together —, so that kernel stays off here)
    O.push(v2_off=1)
    want = [_alone(model, contents, s) for s in seqs]
    pool = SequenceStreams(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), seqs[0], S, pairs=True)
    assert len(pool.models) == 2 * S and len(pool) == S
    lib = _lib.lib()
    try:
        for off in (1, 0):
            O.set(gemm_pair_off=off)
            # stream 0: a pair and an odd one out (solo route); stream 1: a pair
            got = pool.run([[seqs[0], seqs[1], seqs[4]], [seqs[2], seqs[3]]], keep_outputs=True)
            flat = got[0][:2] + got[1] + got[0][2:]
            for k, (g, w) in enumerate(zip(flat, want)):
                assert g.shape == w.shape
                if off:
                    assert torch.equal(g, w), "sequence %d, separate launches" % k
                else:
                    err = float((g - w).abs().max())
                    assert err <= 2e-4 * max(1.0, float(w.abs().max())), "sequence %d: %.3e" % (k, err)
    finally:
        O.pop()
        pool.close()


@pytest.mark.parametrize("group", [3, 4, 8])
def test_lockstep_groups_of_three_and_four(gpu, group):
    """three / four sequences per stream in lock-step (one launch for the products of all of them), a leftover that
    runs solo: bitwise equal to the solo runs with separate launches, float rounding with shared ones"""
    from temporal_latticenet_amd import _lib
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=2, sigma=0.7)
    S = 2
    seqs = [[(torch.from_numpy(p).to(gpu), torch.from_numpy(v).to(gpu)) for p, v in make_sequence(5000 + 400 * s, 2, seed=270 + s)]
            for s in range(2 * group + 1)]
    model = build_model(contents).eval()
    _alone(model, contents, seqs[0])
    randomize_parameters(model, seed=29)
    O.push(v2_off=1)   # (as in the pairs test above)
    want = [_alone(model, contents, s) for s in seqs]
    pool = SequenceStreams(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), seqs[0], S, pairs=group)
    assert len(pool.models) == group * S
    lib = _lib.lib()
    try:
        for off in (1, 0):
            O.set(gemm_pair_off=off)
            got = pool.run([seqs[:group] + seqs[2 * group:], seqs[group:2 * group]], keep_outputs=True)
            flat = got[0][:group] + got[1] + got[0][group:]
            for k, (g, w) in enumerate(zip(flat, want)):
                if off:
                    assert torch.equal(g, w), "sequence %d, separate launches" % k
                else:
                    err = float((g - w).abs().max())
                    assert err <= 2e-4 * max(1.0, float(w.abs().max())), "sequence %d: %.3e" % (k, err)
    finally:
        O.pop()
        pool.close()
