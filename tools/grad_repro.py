"""Run the training-path forward+backward several times on identical inputs and report which parameter gradients
differ between runs (diagnosis of run-to-run nondeterminism)."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from tests.helpers import build_model, make_config, make_lattice, randomize_parameters
from temporal_latticenet_amd.synthetic import make_sequence


def forward(model, contents, seq, grad):
    lat = make_lattice(contents)
    with torch.set_grad_enabled(grad):
        for t, (pos, val) in enumerate(seq):
            logsm, raw, lat = model(lat, torch.from_numpy(pos).cuda(), torch.from_numpy(val).cuda(),
                                    t != len(seq) - 1, grad)
    return logsm, raw


def main():
    rnn = ("linear", "none", "lstm", "maxpool") if len(sys.argv) > 1 and sys.argv[1] == "1" else ("gru", "gru", "aflow", "gru")
    contents = make_config(rnn_modules=rnn, frames=2, sigma=0.8)
    seq = make_sequence(5000, 2, seed=61)
    model = build_model(contents).train()
    with torch.no_grad():
        forward(model, contents, seq, False)
    model.reset_sequence()
    randomize_parameters(model, seed=4)
    target = torch.randint(0, 26, (5000,), generator=torch.Generator().manual_seed(0)).cuda()
    ref = None
    junk = []
    for rep in range(8):
        junk.append(torch.randn(1 << (10 + rep), device="cuda"))      # perturb the allocator between runs
        model.zero_grad(set_to_none=True)
        logsm, raw = forward(model, contents, seq, True)
        model.reset_sequence()
        if rep % 2 == 1:            # as tests/test_gpu_train.py does: an inference pass between forward and backward
            with torch.no_grad():
                forward(model, contents, seq, False)
            model.reset_sequence()
        loss = torch.nn.functional.nll_loss(logsm, target)
        loss.backward()
        got = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        got["__raw__"] = raw.detach().cpu().numpy().copy()
        if ref is None:
            ref = got
            continue
        worst = []
        for k in ref:
            d = float(np.abs(got[k] - ref[k]).max()) / max(float(np.abs(ref[k]).max()), 1e-12)
            if d > 1e-5:
                worst.append((d, k))
        worst.sort(reverse=True)
        print("rep", rep, "loss", float(loss), "params differing >1e-5:", len(worst), worst[:6], flush=True)


if __name__ == "__main__":
    main()
