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
NPTS = int(os.environ.get("TLN_TEST_POINTS", "9000"))
SIGMA = float(os.environ.get("TLN_TEST_SIGMA", "0.7"))
contents = make_config(frames=T, sigma=SIGMA, capacity=1 << 18)
seqs_np = [make_sequence(NPTS, T, seed=50 + s) for s in range(2 if NPTS < 50000 else 1)]
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
    # the stream of sequences as bench.py drives it: key exchange per sequence, one sequence ahead (run_stream)
    outs = runner.run_stream(mine)
    for i, out in enumerate(outs):
        if plan.owns_last_frame():
            assert torch.equal(out[1], want[i]), "frame-sharded logits differ from the sequential run (seq %%d)" %% i
            if os.environ.get("TLN_TEST_ORACLE") == "1":
                # ... and match the CPU oracle (models.py:284-476 restated) at the north-star tolerance
                from tests.helpers import oracle_from_model
                oracle = oracle_from_model(model, contents)
                for t, (p, v) in enumerate(seqs_np[i]):
                    ref = oracle.forward(p, v, early_return=(t != T - 1))
                err = float((out[1].cpu() - ref).abs().max())
                scale = max(1.0, float(ref.abs().max()))
                print("ORACLE max_abs %%.3e max|logit| %%.2f V0 %%d" %% (err, scale, lat.nr_lattice_vertices() if False else runner.lattice.nr_lattice_vertices()))
                assert err <= 1e-4 * scale, (err, scale)
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


@pytest.mark.parametrize("world,frames,route,points", [(2, 2, "program", 9000), (4, 4, "program", 9000),
                                                       (2, 4, "program", 9000), (2, 2, "hooks", 9000),
                                                       (2, 4, "program", 120000)])
def test_frame_sharded_model_equals_sequential(gpu, tmp_path, world, frames, route, points):
    """world == frames: one frame per rank; world < frames: a block of frames per rank (states stay native inside a
    block); route: the native frame program run in segments, or the operator route with forward hooks.  The last case is
    BASELINE config 4's size (4 frames x 120 000 points, sigma 0.6, [gru,gru,aflow,gru]): the frame-sharded logits are
    bitwise the sequential ones AND match the CPU oracle at the north-star tolerance."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TLN_TEST_FRAMES=str(frames), TLN_TEST_ROUTE=route, TLN_TEST_POINTS=str(points),
                   TLN_TEST_SIGMA="0.6" if points >= 50000 else "0.7", TLN_TEST_ORACLE="1" if points >= 50000 else "0")
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
    if points >= 50000:
        line = [ln for o in outs for ln in o.splitlines() if ln.startswith("ORACLE")]
        assert len(line) == 1, outs[-1][-2000:]
        print("[parity] frame-sharded 2 ranks, 4 x 120k: " + line[0])


NCCL_SMOKE = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from temporal_latticenet_amd import dist as D
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rows = torch.arange(21, dtype=torch.int32, device="cuda").reshape(7, 3)
got = D.all_gather_rows(rows)                       # the via_host=False branch: device tensors through RCCL
assert len(got) == 1 and got[0].is_cuda and torch.equal(got[0], rows)
empty = D.all_gather_rows(rows[:0])
assert empty[0].shape == (0, 3)
assert D.max_over_ranks(2.5, device="cuda") == 2.5
dist.barrier()
print("NCCL OK", torch.cuda.get_device_name(0))
dist.destroy_process_group()
'''


def test_rccl_backend_runs_the_collectives_once(gpu, tmp_path):
    """a world of ONE rank over the nccl (= RCCL) backend: the device-tensor branch of the key all-gather and the MAX
    all-reduce of the step time execute on this GPU at least once before the driver's 8-GPU run (point-to-point needs a
    second rank and stays covered by the gloo rehearsals)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "nccl_smoke.py"
    script.write_text(NCCL_SMOKE % {"root": ROOT, "port": str(port)})
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "NCCL OK" in p.stdout, p.stdout[-3000:]
