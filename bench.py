#!/usr/bin/env python3
"""bench.py — point-clouds/sec through the temporal-LatticeNet hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One STEP = one 4-frame sequence (120k points/frame, sigma 0.6, 26 classes, rnn_modules=[gru,gru,aflow,gru], the
reference's pretrained configuration) through LNN_SEQ.forward: distribute -> PointNet pool -> U-Net of lattice
convolutions with GRU/AFlow fusion -> slice, inference mode, inputs already resident in HBM.
Every rank runs its own sequences (a sequence owns its lattice and hidden state: train_ln.py:236-239), so
N GPUs shard the stream of sequences with no data-path collective => "scaling": "weak".

Prints ONE JSON line (rank 0) with `value` = clouds/sec of the whole job, plus
  roofline     : the dominant kernel (k_gather_gemm, fp32 MFMA) timed per launch with HIP events on the launch stream
  cpu_baseline : the CPU oracle (PyTorch eager restatement, kind "port") on a bounded sample, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--rnn", type=str, default="gru,gru,aflow,gru")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-points", type=int, default=120000, help="points per frame of the CPU sample")
    ap.add_argument("--cpu-frames", type=int, default=1)
    ap.add_argument("--breakdown", action="store_true", help="print a per-op time table to stderr")
    return ap.parse_args()


def run_sequence(model, lattice, frames):
    out = None
    for t, (pos, val) in enumerate(frames):
        out, raw, lattice = model(lattice, pos, val, t != len(frames) - 1, False)
    model.reset_sequence()
    return out


def gemm_flops(meta):
    return 2.0 * meta["M"] * meta["N"] * meta["K"]


def gemm_bytes(meta):
    # algorithmic bytes: read every source row once, write the output once, the 9-int table,
    # the weights, the residual if any
    m, n, k = meta["M"], meta["N"], meta["K"]
    b = 4.0 * (m * meta["cin"] + m * n + k * n) + 4.0 * m * meta["taps"]
    if meta["res"]:
        b += 4.0 * m * n
    return b


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from temporal_latticenet_amd import ops
    from temporal_latticenet_amd.configs import build_model, make_config, make_lattice
    from temporal_latticenet_amd.synthetic import make_sequence

    rnn = tuple(args.rnn.split(","))
    contents = make_config(rnn_modules=rnn, frames=args.frames, sigma=args.sigma, capacity=1 << 18)
    devnull = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, devnull          # the model prints its layer list like the reference does
    try:
        torch.manual_seed(1234)
        model = build_model(contents).eval()
    finally:
        sys.stdout = stdout
    lattice = make_lattice(contents)
    seq_np = make_sequence(args.points, args.frames, seed=1234 + rank)
    frames = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in seq_np]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        sys.stdout = devnull
        try:
            run_sequence(model, lattice, frames)       # creates the lazily built parameters
        finally:
            sys.stdout = stdout
        for _ in range(args.warmup):
            run_sequence(model, lattice, frames)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_sequence(model, lattice, frames)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # vertex counts of the workload (data dependent; printed with every result)
        lat = make_lattice(contents)
        vcounts = []
        for t, (pos, val) in enumerate(frames):
            model(lat, pos, val, t != len(frames) - 1, False)
            l1 = lat.coarsen()
            vcounts.append([lat.nr_lattice_vertices(), l1.nr_lattice_vertices(), l1.coarsen().nr_lattice_vertices()])
        model.reset_sequence()

        # ---- roofline pass: per-launch HIP-event timing of the dominant kernel over the same workload ----
        roof = None
        breakdown = {}
        if rank == 0:
            ops.profile_begin()
            reps = max(2, min(args.steps, 5))
            for _ in range(reps):
                run_sequence(model, lattice, frames)
            rec = ops.profile_end()
            for name, ms, meta in rec:
                d = breakdown.setdefault(name, [0, 0.0])
                d[0] += 1
                d[1] += ms
            g = [(ms, meta) for name, ms, meta in rec if name == "gather_gemm"]
            tot_ms = sum(ms for ms, _ in g)
            tot_fl = sum(gemm_flops(m) for _, m in g)
            tot_by = sum(gemm_bytes(m) for _, m in g)
            achieved = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
            roof = {"kernel": "k_gather_gemm", "bound": "mfma", "achieved": round(achieved, 3),
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": None, "launches_per_step": len(g) // reps,
                    "avg_launch_us": round(tot_ms * 1e3 / max(len(g), 1), 2),
                    "flops_per_step": tot_fl / reps, "algorithmic_bytes_per_step": tot_by / reps,
                    "share_of_step_time": round((tot_ms / reps) / (elapsed / args.steps * 1e3), 3)}
            if args.breakdown:
                for k, (cnt, ms) in sorted(breakdown.items(), key=lambda kv: -kv[1][1]):
                    print("  %-16s %5d calls/step %9.3f ms/step" % (k, cnt // reps, ms / reps), file=sys.stderr)

    clouds = args.gpus * args.steps * args.frames
    value = clouds / elapsed

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, contents, args)

    if rank == 0:
        line = {
            "metric": "point-clouds/sec (120k pts, sigma=0.6, 4-frame seq)",
            "value": round(value, 3), "unit": "clouds/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d-frame sequence, %d pts/frame, sigma=%s, rnn_modules=[%s], 26 classes, "
                                   "full U-Net lattice encoder/decoder, inference" % (args.frames, args.points, args.sigma, args.rnn),
                       "parallelism": "one sequence stream per GPU (no data-path collective)",
                       "vertices_per_frame_V0_V1_V2": vcounts},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(model, contents, args):
    """Times the CPU oracle (PyTorch eager restatement of the same path, same weights) on a bounded sample."""
    import torch
    from temporal_latticenet_amd.synthetic import make_sequence
    from oracle.model import OracleLNN
    m = contents["model"]
    cores = min(os.cpu_count() or 1, 32)     # eager ops on a few-thousand-row lattice do not scale past this
    torch.set_num_threads(cores)
    oracle = OracleLNN(model.state_dict(), 26, m["rnn_modules"], m["sequence_learning"], m["pointnet_layers"],
                       m["nr_downsamples"], m["nr_blocks_down_stage"], m["nr_blocks_bottleneck"],
                       m["nr_blocks_up_stage"], [args.sigma] * 3, 1 << 18, m["experiment"])
    seq = make_sequence(args.cpu_points, args.cpu_frames, seed=1234)
    t0 = time.perf_counter()
    for t, (pos, val) in enumerate(seq):
        oracle.forward(pos, val, early_return=(t != len(seq) - 1))
    dt = time.perf_counter() - t0
    return {"value": round(len(seq) / dt, 4), "unit": "clouds/s", "cores": cores, "kind": "port",
            "sample": "%d frames of %d points (same config, frames 0..%d of the sequence), oracle/model.py, %.1f s"
                      % (len(seq), args.cpu_points, len(seq) - 1, dt)}


if __name__ == "__main__":
    main()
