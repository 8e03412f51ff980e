"""GPU: the frame program (temporal_latticenet_amd/engine.py, include/tln.h "Frame program") against the
operator-level route of the same model: identical kernels in identical order, so the tensors must be EQUAL, for
every frame of a sequence including the early-return values (models.py:307, 346, 427)."""
import numpy as np
import pytest
import torch

from tests.helpers import build_model, make_config, make_lattice, oracle_from_model, randomize_parameters
from temporal_latticenet_amd import options as O
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


def _run(model, contents, seq, gpu, switch_off_at=None):
    lat = make_lattice(contents)
    outs, used = [], []
    with torch.no_grad():
        for t, (pos, val) in enumerate(seq):
            if switch_off_at is not None and t == switch_off_at:
                model.use_frame_program = False
            a, b, lat = model(lat, torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), t != len(seq) - 1, False)
            outs.append((a.clone(), b.clone()))
            used.append(bool(getattr(model, "_program_active", False)))
    model.reset_sequence()
    return outs, used


def _prepared(contents, seq, gpu, nr_classes=26, seed=3):
    model = build_model(contents, nr_classes=nr_classes).eval()
    model.use_frame_program = False
    _run(model, contents, seq, gpu)                    # lazily created parameters
    randomize_parameters(model, seed=seed)
    return model


@pytest.mark.parametrize("rnn,seq_learning,frames", [
    (("gru", "gru", "aflow", "gru"), True, 4),         # the pretrained configuration (cfg:59)
    (("aflow", "none", "gru", "aflow"), True, 3),
    (("gru", "none", "none", "none"), True, 3),        # early frames return right after the PointNet (models.py:307)
    (("none", "gru", "none", "none"), True, 3),        # ... after the middle fusion (models.py:346)
    (("gru", "none", "none", "none"), False, 1),       # no sequence learning
    (("linear", "linear", "gru", "linear"), True, 3),  # TemporalLinearModule (lm:149-185)
    (("maxpool", "linear", "lstm", "cga"), True, 3),   # early max-pool fusion (lm:555-563), LSTM, global attention
    (("cga", "lstm", "maxpool", "lstm"), True, 3),
])
def test_program_equals_operator_route(gpu, rnn, seq_learning, frames):
    contents = make_config(rnn_modules=rnn, sequence_learning=seq_learning, frames=frames, sigma=0.7)
    seq = make_sequence(12000, frames, seed=17)
    model = _prepared(contents, seq, gpu)
    ref, used = _run(model, contents, seq, gpu)
    assert not any(used)
    model.use_frame_program = True
    for rep in range(2):                               # twice: hidden states must reset with the sequence
        got, used = _run(model, contents, seq, gpu)
        assert all(used), "the frame program was not used: %s" % (used,)
        for t in range(frames):
            for k in range(2):
                assert got[t][k].shape == ref[t][k].shape
                if k == 0 and t == frames - 1 and got[t][0].shape[1] == 26:
                    # log_softmax(scores): written by the slice head on the program route, torch.nn.LogSoftmax on the
                    # operator route (same formula, not the same instruction sequence)
                    np.testing.assert_allclose(got[t][0].cpu().numpy(), ref[t][0].cpu().numpy(), rtol=2e-6, atol=2e-6)
                    continue
                assert torch.equal(got[t][k], ref[t][k]), "frame %d output %d differs (max %.3e)" % (
                    t, k, float((got[t][k] - ref[t][k]).abs().max()))


def test_program_matches_oracle(gpu):
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=3, sigma=0.6)
    seq = make_sequence(15000, 3, seed=31)
    model = _prepared(contents, seq, gpu, seed=1)
    model.use_frame_program = True
    got, used = _run(model, contents, seq, gpu)
    assert all(used)
    oracle = oracle_from_model(model, contents)
    for t, (pos, val) in enumerate(seq):
        want = oracle.forward(pos, val, early_return=(t != len(seq) - 1))
        g = got[t][1].cpu()
        assert g.shape == want.shape
        err = float((g - want).abs().max())
        assert err <= 1e-4 * max(1.0, float(want.abs().max())), "frame %d: %.3e" % (t, err)


def test_early_return_values_are_optional(gpu):
    """keep_early_values = False: an early-return frame hands back None (the value the reference's loops drop is not
    copied out of the program's state buffer, tln_program_run with d_out = NULL) and the last frame's outputs — one
    sequence alone and a lock-step group — do not change by a bit"""
    from temporal_latticenet_amd.models import forward_group
    from temporal_latticenet_amd.streams import share_parameters
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=3, sigma=0.7)
    seq = make_sequence(9000, 3, seed=23)
    model = _prepared(contents, seq, gpu)
    model.use_frame_program = True
    ref, used = _run(model, contents, seq, gpu)
    assert all(used)
    model.keep_early_values = False
    lat = make_lattice(contents)
    with torch.no_grad():
        for t, (pos, val) in enumerate(seq):
            a, b, lat = model(lat, torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), t != 2, False)
            if t != 2:
                assert a is None and b is None
    model.reset_sequence()
    assert torch.equal(b, ref[2][1]) and torch.equal(a, ref[2][0])
    # a lock-step group of two (the second one a replica sharing the parameters)
    other = build_model(contents).eval()
    other.use_frame_program = False
    _run(other, contents, seq, gpu)
    other = share_parameters(other, model)
    other.use_frame_program = True
    kept = None
    for keep in (True, False):
        model.keep_early_values = other.keep_early_values = keep
        lats = [make_lattice(contents), make_lattice(contents)]
        with torch.no_grad():
            for t, (pos, val) in enumerate(seq):
                p, v = torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu)
                res = forward_group([model, other], lats, [p, p], [v, v], t != 2)
                lats = [r[2] for r in res]
                if t != 2:
                    assert all((r[1] is None) == (not keep) for r in res)
        model.reset_sequence()
        other.reset_sequence()
        if keep:
            kept = [r[1].clone() for r in res]
        else:
            for r, w in zip(res, kept):
                assert torch.equal(r[1], w)


def test_hidden_states_follow_a_frame_that_needs_the_operator_route(gpu):
    """frames 0-1 through the program, frames 2-3 through the modules: the states are handed over"""
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=4, sigma=0.7)
    seq = make_sequence(9000, 4, seed=5)
    model = _prepared(contents, seq, gpu)
    ref, _ = _run(model, contents, seq, gpu)
    model.use_frame_program = True
    got, used = _run(model, contents, seq, gpu, switch_off_at=2)
    assert used == [True, True, False, False]
    for t in range(4):
        assert torch.equal(got[t][1], ref[t][1]), "frame %d" % t


def test_unsupported_configuration_stays_on_the_operator_route(gpu):
    contents = make_config(rnn_modules=("none", "gru", "none", "gru"), frames=2, sigma=0.8,
                           experiment="pointnet_no_elevate")
    seq = make_sequence(6000, 2, seed=6)
    model = _prepared(contents, seq, gpu)
    model.use_frame_program = True
    _, used = _run(model, contents, seq, gpu)
    assert not any(used)


def test_program_is_rebuilt_after_a_parameter_update(gpu):
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=2, sigma=0.8)
    seq = make_sequence(6000, 2, seed=8)
    model = _prepared(contents, seq, gpu)
    model.use_frame_program = True
    a, _ = _run(model, contents, seq, gpu)
    with torch.no_grad():
        model.recurrent_fusion_modules[1].AFLOW.alpha.fill_(0.25)      # baked into the program by value
        model.slice_fast_cuda.linear_clasify.bias.add_(1.0)
    b, used = _run(model, contents, seq, gpu)
    assert all(used)
    model.use_frame_program = False
    c, _ = _run(model, contents, seq, gpu)
    assert torch.equal(b[-1][1], c[-1][1])
    assert not torch.equal(a[-1][1], b[-1][1])


@pytest.mark.parametrize("n,sigma,capacity", [(777, 2.5, 100000), (3001, 0.4, 600), (64, 5.0, 100000)])
def test_program_edge_cases_equal_operator_route(gpu, n, sigma, capacity):
    """ragged sizes: a handful of vertices (fewer than one 32-row tile), a capacity that rejects most keys (rows with
    index -1, models.py:479), a cloud of one wave"""
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=3, sigma=sigma, capacity=capacity)
    seq = make_sequence(n, 3, seed=23)
    model = _prepared(contents, seq, gpu)
    ref, _ = _run(model, contents, seq, gpu)
    model.use_frame_program = True
    got, used = _run(model, contents, seq, gpu)
    assert all(used)
    for t in range(3):
        a, b = got[t][1], ref[t][1]
        assert a.shape == b.shape
        same = torch.equal(a, b) or bool(((a == b) | (a.isnan() & b.isnan())).all())   # AFlow's 0/0 is a NaN on both
        assert same, "frame %d differs (max %.3e)" % (t, float((a - b).abs().nan_to_num().max()))


@pytest.mark.parametrize("rnn,seq_learning,frames", [
    (("gru", "gru", "aflow", "gru"), True, 3),
    (("gru", "none", "none", "none"), True, 3),        # early frames stop inside the level-0 prefix
    (("none", "gru", "none", "none"), True, 2),
    (("gru", "none", "none", "none"), False, 1),       # no sequence learning: every frame resets the lattice
    (("maxpool", "linear", "lstm", "cga"), True, 3),
])
def test_lockstep_pair_equals_the_solo_frames(gpu, rnn, seq_learning, frames):
    """models.forward_pair on two sequences of very different size: every frame's outputs (early-return values and
    final scores) bitwise equal to the solo frame program when the paired products are issued as two launches, and
    within float rounding of it when they share a launch; hidden states reset with the sequences"""
    from temporal_latticenet_amd import _lib
    from temporal_latticenet_amd.models import forward_pair
    from temporal_latticenet_amd.streams import share_parameters
    contents = make_config(rnn_modules=rnn, sequence_learning=seq_learning, frames=frames, sigma=0.7)
    seqs = [make_sequence(12000, frames, seed=31), make_sequence(2500, frames, seed=32)]
    model = _prepared(contents, seqs[0], gpu)
    model.use_frame_program = True
    want = [_run(model, contents, s, gpu)[0] for s in seqs]
    twin = build_model(contents).eval()
    _run(twin, contents, seqs[0], gpu)
    twin = share_parameters(twin, model)
    twin.use_frame_program = True
    models = [model, twin]
    lib = _lib.lib()
    # the bitwise half of the statement needs the same kernel on both routes: products that share a launch may take the
    # large-M kernel where each alone takes the direct one (their rows are counted together), so that kernel stays off
    # here; tests/test_gpu_fullsize.py runs the lock-step route with it against the oracle
    O.push(v2_off=1)
    want = [_run(model, contents, s, gpu)[0] for s in seqs]
    try:
        for off in (1, 0, 0):
            O.set(gemm_pair_off=off)
            lats = [make_lattice(contents), make_lattice(contents)]
            with torch.no_grad():
                for t in range(frames):
                    pos = [torch.from_numpy(s[t][0]).to(gpu) for s in seqs]
                    val = [torch.from_numpy(s[t][1]).to(gpu) for s in seqs]
                    res = forward_pair(models, lats, pos, val, t != frames - 1)
                    assert all(getattr(m, "_program_active", False) for m in models)
                    for k in range(2):
                        for j in range(2):
                            g, w = res[k][j], want[k][t][j]
                            assert g.shape == w.shape
                            if off:
                                assert torch.equal(g, w), "sequence %d frame %d output %d" % (k, t, j)
                            else:
                                err = float((g - w).abs().max())
                                assert err <= 2e-4 * max(1.0, float(w.abs().max())), (k, t, j, err)
                        lats[k] = res[k][2]
            for m in models:
                m.reset_sequence()
    finally:
        O.pop()


def test_slice_head_writes_log_softmax(gpu):
    """models.py:466-468 returns (log_softmax(scores), scores): on the frame-program route the slice head writes both
    (no torch kernel in the frame); the operator route applies torch.nn.LogSoftmax as the reference does"""
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=2, sigma=0.8)
    seq = make_sequence(9000, 2, seed=12)
    model = _prepared(contents, seq, gpu)
    model.use_frame_program = True
    lat = make_lattice(contents)
    with torch.no_grad():
        for t, (pos, val) in enumerate(seq):
            logsm, raw, lat = model(lat, torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), t != len(seq) - 1, False)
    assert model._program.fused_logsm and model._program_active
    model.reset_sequence()
    want = torch.log_softmax(raw, 1)
    assert logsm.shape == raw.shape == (9000, 26)
    np.testing.assert_allclose(logsm.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(torch.exp(logsm).sum(1).cpu().numpy(), 1.0, rtol=0, atol=1e-5)


def test_a_frame_whose_coarse_levels_outgrow_the_planning_prediction_is_planned_again(gpu):
    """The frame program sizes its arena and every coarse-level buffer before the coarse levels' vertex counts have reached
    the host, with a prediction (old count + max(2048, r x new level-0 vertices), program.hip::predict_bounds) instead of
    the lattice's hard bound (4 / 16 x): ~0.3 GB per resident sequence instead of 1.9.  A cloud of ISOLATED points (every
    fine vertex far from the next: up to four coarse vertices each) breaks the prediction: the frame is planned again with
    the exact counts (replan: arena regrown with its contents kept, walk state replayed) and gives the operator route's
    bits; so does the lock-step group."""
    from temporal_latticenet_amd.models import forward_group
    from temporal_latticenet_amd.streams import share_parameters
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=2, sigma=0.05, capacity=1 << 17)
    rng = np.random.default_rng(3)
    seq = []
    for t in range(2):              # 6000 points scattered over (60 m)^3: no two share a lattice cell at sigma 0.05
        pos = (rng.uniform(-30, 30, (6000, 3))).astype(np.float32)
        seq.append((pos, rng.uniform(0, 1, (6000, 1)).astype(np.float32)))
    model = build_model(contents).eval()
    _run(model, contents, seq, gpu)
    randomize_parameters(model, seed=23)
    model.use_frame_program = False
    want, _ = _run(model, contents, seq, gpu)
    model.use_frame_program = True
    got, used = _run(model, contents, seq, gpu)
    assert all(used)
    prog = model._program
    assert prog is not None and prog.replans() >= 1, "the fixture was meant to break the prediction"
    lat = make_lattice(contents)
    with torch.no_grad():
        model(lat, torch.from_numpy(seq[0][0]).to(gpu), torch.from_numpy(seq[0][1]).to(gpu), True, False)
    l1 = lat.coarsen()
    v0, v1, v2 = lat.nr_lattice_vertices(), l1.nr_lattice_vertices(), l1.coarsen().nr_lattice_vertices()
    model.reset_sequence()
    # more coarse vertices than the prediction allows for (old count + max(2048, r x new level-0 vertices), r = 1, 1/2)
    assert v1 > v0 + 2048 or v2 > v0 // 2 + 2048, (v0, v1, v2)
    for t in range(2):
        for j in range(2):
            if t == 1 and j == 0:       # (the program's slice head writes its own log-softmax: torch's differs in the last bits)
                assert float((got[t][j] - want[t][j]).abs().max()) < 1e-5
            else:
                assert torch.equal(got[t][j], want[t][j]), "frame %d output %d" % (t, j)
    # the same through a lock-step group of two
    twin = build_model(contents).eval()
    _run(twin, contents, seq, gpu)
    twin = share_parameters(twin, model)
    before = prog.replans()
    prog.apply_options()
    lats = [make_lattice(contents), make_lattice(contents)]
    with torch.no_grad():
        for t in range(2):
            pos = [torch.from_numpy(seq[t][0]).to(gpu)] * 2
            val = [torch.from_numpy(seq[t][1]).to(gpu)] * 2
            res = forward_group([model, twin], lats, pos, val, t != 1)
            for k in range(2):
                assert torch.equal(res[k][1], want[t][1]), "group member %d frame %d" % (k, t)
                lats[k] = res[k][2]
    for m in (model, twin):
        m.reset_sequence()
    assert model._program.replans() > before - 1


def test_memory_of_a_resident_sequence(gpu):
    """~2 GB per resident sequence for a working set of 0.15 GB.  With the capacity taken from the cloud size
    (configs.suggest_capacity), the pool's accumulators sized by the vertices the level holds and the frame planned with
    predicted coarse counts: <= 0.6 GB per 4 x 120k sequence, by owner (tln_lattice_memory / tln_program_memory)."""
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=4, sigma=0.6, capacity="auto")
    seq = make_sequence(120000, 4, seed=1234)
    model = build_model(contents).eval()
    for rep in range(2):            # (the first sequence creates the lazily built parameters on the operator route)
        lat = make_lattice(contents, nr_points=120000, frames=4)
        assert lat.capacity() <= 110000
        with torch.no_grad():
            for t, (p, v) in enumerate(seq):
                a, b, lat = model(lat, torch.from_numpy(p).to(gpu), torch.from_numpy(v).to(gpu), t != 3, False)
        if rep == 0:
            model.reset_sequence()
    assert model._program is not None and model._program_active
    lm, pm = lat.memory_bytes(), model._program.memory_bytes()
    model.reset_sequence()
    print("[memory] lattice %s" % {k: round(v / 2 ** 20, 1) for k, v in lm.items()})
    print("[memory] program %s replans %d" % ({k: round(v / 2 ** 20, 1) for k, v in pm.items()}, model._program.replans()))
    assert lat.overflow_rows() == 0 and lat.nr_lattice_vertices() > 25000
    assert model._program.replans() == 0
    assert lm["total"] + pm["total"] <= 600 * 2 ** 20, (lm, pm)
    assert pm["arena"] <= 2.0 * pm["arena_high_water"] + 2 ** 24
