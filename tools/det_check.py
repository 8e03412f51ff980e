import sys, warnings; sys.path.insert(0, ".")
import torch
from tests.helpers import build_model, make_config, randomize_parameters
from temporal_latticenet_amd.synthetic import make_sequence
from tools.grad_repro import forward
torch.manual_seed(20240607)
contents = make_config(frames=2, sigma=0.8)
seq = make_sequence(6000, 2, seed=63)
model = build_model(contents).train()
with torch.no_grad():
    forward(model, contents, seq, False)
model.reset_sequence()
randomize_parameters(model, seed=6)
target = torch.randint(0, 26, (6000,), generator=torch.Generator().manual_seed(1)).cuda()
torch.use_deterministic_algorithms(True, warn_only=True)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    logsm, _ = forward(model, contents, seq, True)
    torch.nn.functional.nll_loss(logsm, target).backward()
msgs = sorted(set(str(x.message)[:160] for x in w))
print("\n".join(msgs))
