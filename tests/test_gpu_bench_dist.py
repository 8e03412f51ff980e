"""GPU: bench.py itself as the driver launches it for N > 1 — one process per rank, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment — rehearsed on the one GPU of the test box over gloo (`--dist-backend gloo --same-device`):
sequence sharding with 2 ranks, frame sharding (key all-gather + hidden-state hand-off, temporal_latticenet_amd/dist.py)
with 2 and 4 ranks.  Every run must end with ONE JSON line on rank 0 that carries the contract's fields."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--points", "6000", "--dist-backend", "gloo", "--same-device", "--no-cpu-baseline"] + extra
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, e[-3000:])
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0][-2000:]
    for o, _ in outs[1:]:
        assert not any(ln.startswith("{") for ln in o.splitlines()), "only rank 0 prints"
    return json.loads(lines[0])


def _check_line(d, world):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_scatter", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "clouds/s"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert "6k pts" in d["metric"] and "workload" in d["config"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["roofline_scatter"]["bound"] == "hbm" and set(d["roofline_scatter"]["stages"]) >= {"K1_distribute",
                                                                                              "K2_pointnet_pool"}


def test_bench_sequence_sharding_two_ranks(gpu):
    d = _launch(2, ["--streams", "2", "--pairs", "2"])
    _check_line(d, 2)
    # value = clouds of ALL ranks / time: 2 ranks x 2 streams x 2 lock-stepped sequences x 2 steps x 4 frames
    assert abs(d["value"] - 2 * 2 * 2 * 2 * 4 / (d["ms_per_step"] * 2e-3)) < 1e-3 * d["value"]
    assert "2 lock-stepped" in d["config"]["parallelism"]
    assert "no data-path collective" in d["config"]["parallelism"]
    # ... and the same invocation also times the north star's frame sharding over the two ranks (4 frames: blocks of two)
    fm = d["frames_mode"]
    assert fm["ranks_per_sequence"] == 2 and fm["groups"] == 1
    assert fm["latency_ms_per_sequence"] > 0 and fm["steady_state_clouds_per_s"] > 0


@pytest.mark.parametrize("world", [2, 4])
def test_bench_frame_sharding(gpu, world):
    d = _launch(world, ["--mode", "frames"])
    _check_line(d, world)
    assert "sharded over %d ranks" % world in d["config"]["parallelism"]
    assert d["frames_mode"]["latency_ms_per_sequence"] > 0 and d["frames_mode"]["steady_state_clouds_per_s"] > 0


def test_bench_under_torch_distributed_run(gpu):
    """exactly the driver's launch line for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — with two ranks on the one GPU over gloo: the elastic
    agent hosts the rendezvous store of the job, the children of `--frames-extra` must host their own"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--points", "6000", "--dist-backend", "gloo", "--same-device", "--no-cpu-baseline", "--streams", "2", "--pairs", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    _check_line(d, 2)
    fm = d["frames_mode"]
    assert "error" not in fm, fm
    assert fm["ranks_per_sequence"] == 2 and fm["steady_state_clouds_per_s"] > 0 and fm["backend"] == "gloo"
