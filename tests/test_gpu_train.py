"""GPU: the training path.  loss.backward() through the HIP forward must give the same
parameter gradients as PyTorch autograd through the CPU oracle, and an optimiser step must work the way
train_ln.py:212-233 drives it (loss on the last frame only, AdamW)."""
import numpy as np
import pytest
import torch

from tests.helpers import build_model, make_config, make_lattice, oracle_from_model, randomize_parameters
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


def _forward(model, contents, seq, gpu, grad):
    lat = make_lattice(contents)
    with torch.set_grad_enabled(grad):
        for t, (pos, val) in enumerate(seq):
            logsm, raw, lat = model(lat, torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu),
                                    t != len(seq) - 1, grad)
    return logsm, raw


@pytest.mark.parametrize("rnn", [("gru", "gru", "aflow", "gru"), ("linear", "none", "lstm", "maxpool")])
def test_parameter_gradients_match_oracle_autograd(gpu, rnn):
    contents = make_config(rnn_modules=rnn, frames=2, sigma=0.8)
    seq = make_sequence(5000, 2, seed=61)
    model = build_model(contents).train()
    with torch.no_grad():
        _forward(model, contents, seq, gpu, False)
    model.reset_sequence()
    randomize_parameters(model, seed=4)
    target = torch.randint(0, 26, (5000,), generator=torch.Generator().manual_seed(0))

    # training-path forward == fused inference forward
    logsm, raw = _forward(model, contents, seq, gpu, True)
    model.reset_sequence()
    with torch.no_grad():
        _, raw_inf = _forward(model, contents, seq, gpu, False)
    model.reset_sequence()
    np.testing.assert_allclose(raw.detach().cpu().numpy(), raw_inf.cpu().numpy(), rtol=1e-4, atol=1e-4)

    loss = torch.nn.functional.nll_loss(logsm, target.to(gpu))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}

    oracle = oracle_from_model(model, contents, dtype=torch.float64)     # float64: the reference carries no rounding noise
    oracle.exact_pool = "grad"           # arg-max rows chosen in the pinned fma order (the rows the HIP pool chooses), the
                                         # differentiated values F.linear's of those rows
    for v in oracle.sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    for t, (pos, val) in enumerate(seq):
        sv = oracle.forward(pos, val, early_return=(t != len(seq) - 1))
    oloss = torch.nn.functional.nll_loss(torch.log_softmax(sv, 1), target)
    oloss.backward()
    assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-4
    checked, worst, bad = 0, 0.0, []
    for k, g in got.items():
        og = oracle.sd[k].grad
        og = None if og is None else og.float()
        if og is None:
            assert float(g.abs().max()) == 0.0, k       # e.g. the never-used AFLOW.weight (lm:291)
            continue
        scale = max(float(og.abs().max()), 1e-6)
        err_max = float((g - og).abs().max()) / scale
        err_l2 = float((g - og).norm()) / max(float(og.norm()), 1e-9)
        # Against a FLOAT64 reference (round 2 compared with a float32 CPU pass at 5e-3).  What remains is the float32
        # rounding of the GPU pass itself, and it depends on the weights: 1.5e-4 median / 8e-4 worst for one draw, 1.1e-3 /
        # 1.9e-3 for this test's seed (deepest layers worst; tools/grad_check.py) -- identical to three digits whether the
        # products' backward runs on the HIP kernels or on the torch formulation, i.e. it comes from the float32 torch ops
        # both share (GroupNorm, the GRU / AFlow arithmetic).  The kernels themselves are pinned against the torch
        # formulation at 2e-4 and against float64 products at 2e-5 in the two tests below.
        tol = 5e-3 if k.startswith("point_net_seq.layers.") else 2.5e-3
        if og.numel() == 1:
            tol = 1e-2          # AFlow's alpha / beta: ONE number summed over every vertex and tap with cancellation
        if err_l2 > 5e-4:
            print("[gradients] %-60s l2 %.3e max %.3e" % (k, err_l2, err_max))
        bad += [(k, err_l2, err_max)] if not (err_l2 < tol and err_max < 10 * tol) else []
        worst = max(worst, err_l2)
        checked += 1
    assert not bad, bad
    assert checked > 40
    print("[gradients] %s: %d parameters, worst relative l2 error %.3e" % (",".join(rnn), checked, worst))
    # every parameter the oracle gives a gradient to also got one on the GPU
    for k, v in oracle.sd.items():
        if v.is_floating_point() and v.grad is not None and float(v.grad.abs().max()) > 0:
            assert k in got, k


def test_training_step_reduces_the_loss(gpu):
    contents = make_config(frames=2, sigma=0.8)
    seq = make_sequence(4000, 2, seed=62)
    model = build_model(contents).train()
    with torch.no_grad():
        _forward(model, contents, seq, gpu, False)
    model.reset_sequence()
    target = (torch.from_numpy(seq[-1][0][:, 0]) > 0).long().to(gpu) + 1      # a learnable 2-class target
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-3, amsgrad=True)   # train_ln.py:181
    losses = []
    for it in range(6):
        logsm, _ = _forward(model, contents, seq, gpu, True)
        loss = torch.nn.functional.nll_loss(logsm, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        model.reset_sequence()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses


def test_backward_kernels_are_deterministic_and_agree_with_the_torch_formulation(gpu):
    """csrc/backward.hip: dW by MFMA tiles with a fixed slice order, dA as a gather-GEMM through the paired taps, the
    slice blends' backward as segment sums — (i) two backward passes give the SAME BITS for every parameter (the torch
    formulation scatters with index_add_: float atomics), (ii) the gradients agree with that formulation (materialised
    im2row + index_add_, autograd.torch_backward) to float rounding."""
    from temporal_latticenet_amd import autograd as AG
    contents = make_config(frames=2, sigma=0.8)
    seq = make_sequence(6000, 2, seed=63)
    model = build_model(contents).train()
    with torch.no_grad():
        _forward(model, contents, seq, gpu, False)
    model.reset_sequence()
    randomize_parameters(model, seed=6)
    target = torch.randint(0, 26, (6000,), generator=torch.Generator().manual_seed(1)).to(gpu)

    def grads():
        model.zero_grad(set_to_none=True)
        logsm, _ = _forward(model, contents, seq, gpu, True)
        torch.nn.functional.nll_loss(logsm, target).backward()
        model.reset_sequence()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    a, b = grads(), grads()
    assert a.keys() == b.keys() and len(a) > 40
    differ = [k for k in a if not torch.equal(a[k], b[k])]
    assert not differ, "two backward passes differ in %d of %d parameters: %s" % (len(differ), len(a), differ)
    AG.torch_backward(True)
    try:
        ref = grads()
    finally:
        AG.torch_backward(False)
    for k in a:
        scale = max(float(ref[k].abs().max()), 1e-9)
        assert float((a[k] - ref[k]).abs().max()) <= 2e-4 * scale + 1e-7, k


@pytest.mark.parametrize("cin,cout,taps", [(64, 64, 9), (192, 192, 9), (128, 64, 9), (192, 576, 1), (96, 8, 1), (192, 26, 1)])
def test_gather_gemm_dw_kernel(gpu, cin, cout, taps):
    """tln_gather_gemm_dw against im2row(src)^T @ dout in float64, on a ragged ~9k-vertex lattice"""
    from oracle import ops as O
    from oracle import permuto as P
    from temporal_latticenet_amd import ops
    from temporal_latticenet_amd.lattice import Lattice
    pos, val = make_sequence(20000, 1, seed=21)[0]
    lat = Lattice.from_params([0.5] * 3, 1 << 17)
    lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu))
    V = lat.nr_lattice_vertices()
    tab = lat.neighbour_table().cpu().numpy()
    g = torch.Generator().manual_seed(cin + cout)
    src = torch.randn(V, cin, generator=g)
    dout = torch.randn(V, cout, generator=g)
    dw = ops.gather_gemm_dw(src.to(gpu), lat.neighbour_table_ptr() if taps == 9 else None, taps, dout.to(gpu), V)
    again = ops.gather_gemm_dw(src.to(gpu), lat.neighbour_table_ptr() if taps == 9 else None, taps, dout.to(gpu), V)
    assert torch.equal(dw, again)
    a = (O.im2row(src, tab) if taps == 9 else src).double()
    want = a.t() @ dout.double()
    scale = float(want.abs().max())
    assert tuple(dw.shape) == (taps * cin, cout)
    assert float((dw.cpu().double() - want).abs().max()) <= 2e-5 * scale
