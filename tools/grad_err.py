"""Per-parameter gradient error (GPU training path vs CPU oracle autograd), for calibrating tests/test_gpu_train.py."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from tests.helpers import build_model, make_config, make_lattice, oracle_from_model, randomize_parameters
from temporal_latticenet_amd.synthetic import make_sequence
from tools.grad_repro import forward

for rnn in [("gru", "gru", "aflow", "gru"), ("linear", "none", "lstm", "maxpool")]:
    contents = make_config(rnn_modules=rnn, frames=2, sigma=0.8)
    seq = make_sequence(5000, 2, seed=61)
    model = build_model(contents).train()
    with torch.no_grad():
        forward(model, contents, seq, False)
    model.reset_sequence()
    randomize_parameters(model, seed=4)
    target = torch.randint(0, 26, (5000,), generator=torch.Generator().manual_seed(0))
    logsm, raw = forward(model, contents, seq, True)
    model.reset_sequence()
    loss = torch.nn.functional.nll_loss(logsm, target.cuda())
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    for rep in range(3):
        oracle = oracle_from_model(model, contents)
        for v in oracle.sd.values():
            if v.is_floating_point():
                v.requires_grad_(True)
        for t, (pos, val) in enumerate(seq):
            sv = oracle.forward(pos, val, early_return=(t != len(seq) - 1))
        oloss = torch.nn.functional.nll_loss(torch.log_softmax(sv, 1), target)
        oloss.backward()
        errs = []
        for k, g in got.items():
            og = oracle.sd[k].grad
            if og is None:
                continue
            errs.append((float((g - og).norm()) / max(float(og.norm()), 1e-9), k))
        errs.sort(reverse=True)
        print(rnn[0], "rep", rep, "threads", torch.get_num_threads(), ["%.2e %s" % e for e in errs[:5]], "median %.2e" % np.median([e[0] for e in errs]), flush=True)
