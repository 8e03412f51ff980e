"""GPU: the frame-sharded multi-rank path on the real HIP model.  Two ranks share the one GPU of the test box and
talk over gloo (host-staged); the driver's multi-GPU runs use the same code over RCCL.  The frame-sharded logits must
be BITWISE equal to the single-process sequential run (deterministic kernels, identical vertex numbering)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, io, contextlib
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from temporal_latticenet_amd import dist as D
from temporal_latticenet_amd.configs import build_model, make_config, make_lattice
from temporal_latticenet_amd.synthetic import make_sequence
from tests.helpers import randomize_parameters

rank, world = D.init_from_env("gloo")
torch.cuda.set_device(0)
T = int(os.environ["TLN_TEST_FRAMES"])
use_program = os.environ["TLN_TEST_ROUTE"] == "program"
contents = make_config(frames=T, sigma=0.7)
seqs_np = [make_sequence(9000, T, seed=50 + s) for s in range(2)]
seqs = [[(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in s] for s in seqs_np]
with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(7)
    model = build_model(contents).eval()
    # lazily created parameters + identical weights on both ranks
    lat = make_lattice(contents)
    with torch.no_grad():
        for t, (p, v) in enumerate(seqs[0]):
            model(lat, p, v, t != T - 1, False)
    model.reset_sequence()
    randomize_parameters(model, 5)

def sequential(seq):
    lat = make_lattice(contents)
    with torch.no_grad():
        for t, (p, v) in enumerate(seq):
            a, b, lat = model(lat, p, v, t != T - 1, False)
    model.reset_sequence()
    return b.clone()

want = [sequential(s) for s in seqs]
plan = D.FrameShardPlan(T, rank, world)
runner = D.FrameShardRunner(model, lambda: make_lattice(contents), plan, via_host=True, use_program=use_program)
mine = [{f: s[f] for f in plan.frames} for s in seqs]
with torch.no_grad():
    keys = runner.exchange_keys(mine)
    for i, frames in enumerate(mine):
        out = runner.run_sequence(frames, keys[i])
        if plan.owns_last_frame():
            assert torch.equal(out[1], want[i]), "frame-sharded logits differ from the sequential run (seq %%d)" %% i
if use_program:
    assert not runner._hooked and getattr(model, "_program", None) is not None, "the frame program was not used"
else:
    assert runner._hooked
runner.close()
assert model.use_frame_program
dist.barrier()
print("RANK %%d OK" %% rank)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,frames,route", [(2, 2, "program"), (4, 4, "program"), (2, 4, "program"), (2, 2, "hooks")])
def test_frame_sharded_model_equals_sequential(gpu, tmp_path, world, frames, route):
    """world == frames: one frame per rank; world < frames: a block of frames per rank (states stay native inside a
    block); route: the native frame program run in segments, or the operator route with forward hooks"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TLN_TEST_FRAMES=str(frames), TLN_TEST_ROUTE=route)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("RANK %d OK" % r) in o, o[-3000:]
