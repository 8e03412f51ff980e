import sys
sys.path.insert(0, ".")
import numpy as np, torch
from tests.helpers import build_model, make_config, make_lattice, oracle_from_model, randomize_parameters
from temporal_latticenet_amd.synthetic import make_sequence
from temporal_latticenet_amd import autograd as AG
from tools.grad_repro import forward
rnn = ("gru", "gru", "aflow", "gru")
contents = make_config(rnn_modules=rnn, frames=2, sigma=0.8)
seq = make_sequence(5000, 2, seed=61)
model = build_model(contents).train()
with torch.no_grad():
    forward(model, contents, seq, False)
model.reset_sequence()
randomize_parameters(model, seed=4)
target = torch.randint(0, 26, (5000,), generator=torch.Generator().manual_seed(0))
oracle = oracle_from_model(model, contents, dtype=torch.float64)
oracle.exact_pool = "grad"
for v in oracle.sd.values():
    if v.is_floating_point():
        v.requires_grad_(True)
for t, (pos, val) in enumerate(seq):
    sv = oracle.forward(pos, val, early_return=(t != len(seq) - 1))
torch.nn.functional.nll_loss(torch.log_softmax(sv, 1), target).backward()
for torch_bw in (False, True):
    for between in (False, True):
        AG.torch_backward(torch_bw)
        model.zero_grad(set_to_none=True)
        logsm, raw = forward(model, contents, seq, True)
        model.reset_sequence()
        if between:
            with torch.no_grad():
                forward(model, contents, seq, False)
            model.reset_sequence()
        torch.nn.functional.nll_loss(logsm, target.cuda()).backward()
        errs = []
        for k, p in model.named_parameters():
            og = oracle.sd[k].grad
            if p.grad is None or og is None:
                continue
            g = p.grad.detach().cpu().double()
            errs.append((float((g - og).norm()) / max(float(og.norm()), 1e-12), k))
        errs.sort(reverse=True)
        print("torch_backward=%s inference_between=%s: worst %s  median %.2e" % (torch_bw, between, ["%.2e %s" % e for e in errs[:3]], np.median([e[0] for e in errs])), flush=True)
AG.torch_backward(False)
