#!/usr/bin/env python3
"""bench.py — point-clouds/sec through the temporal-LatticeNet hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One STEP = one 4-frame sequence (120k points/frame, sigma 0.6, 26 classes, rnn_modules=[gru,gru,aflow,gru], the
reference's pretrained configuration) through LNN_SEQ.forward: distribute -> PointNet pool -> U-Net of lattice
convolutions with GRU/AFlow fusion -> slice, inference mode, inputs already resident in HBM.

--mode sequences (default): every rank runs its own sequences (a sequence owns its lattice and hidden state,
    train_ln.py:236-239), so N GPUs shard the stream of sequences with no data-path collective.
--mode frames: the frames of every sequence are sharded over the ranks of a group (rank g owns frame-slot g):
    all-gather of the per-frame vertex keys, point-to-point hand-off of the fusion modules' hidden states
    (temporal_latticenet_amd/dist.py); with N > frames the N/frames groups take different sequences.
Both are "scaling": "weak" (fixed work per GPU).

Prints ONE JSON line (rank 0) with `value` = clouds/sec of the whole job, plus
  roofline     : the dominant kernel (k_gather_gemm, fp32 MFMA) timed per launch with HIP events on the launch stream
  cpu_baseline : the CPU oracle (PyTorch eager restatement, kind "port") on a bounded sample, rank 0 / N=1 only.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--rnn", type=str, default="gru,gru,aflow,gru")
    ap.add_argument("--mode", choices=["sequences", "frames"], default="sequences")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on the GPU node; gloo for rehearsals")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: all ranks use cuda:0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-points", type=int, default=120000, help="points per frame of the CPU sample")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="lower bound of CPU work in the sample")
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
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", local_rank)} if args.dist_backend == "nccl" else {}
        dist.init_process_group(args.dist_backend, **kw)
    via_host = args.dist_backend != "nccl"

    from temporal_latticenet_amd import dist as D
    from temporal_latticenet_amd import ops
    from temporal_latticenet_amd.configs import build_model, make_config, make_lattice
    from temporal_latticenet_amd.synthetic import make_sequence

    rnn = tuple(args.rnn.split(","))
    contents = make_config(rnn_modules=rnn, frames=args.frames, sigma=args.sigma, capacity=1 << 18)
    quiet = contextlib.redirect_stdout(io.StringIO())   # the model prints its layer list like the reference does
    with quiet:
        torch.manual_seed(1234)
        model = build_model(contents).eval()
    lattice = make_lattice(contents)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    frames_mode = args.mode == "frames" and world > 1
    if frames_mode:
        plan = D.FrameShardPlan(args.frames, rank, world)
        group = None
        if plan.nr_groups > 1:
            for gi in range(plan.nr_groups):
                gr = dist.new_group(list(range(gi * plan.group_size, (gi + 1) * plan.group_size)))
                if gi == plan.group:
                    group = gr
        seed = 1234 + plan.group
    else:
        plan, group, seed = None, None, 1234 + rank
    seq_np = make_sequence(args.points, args.frames, seed=seed)
    frames = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in seq_np]

    with torch.no_grad():
        with quiet:
            run_sequence(model, lattice, frames)       # creates the lazily built parameters (same seed on all ranks)
        if frames_mode:
            runner = D.FrameShardRunner(model, lambda: make_lattice(contents), plan, group=group, via_host=via_host)
            mine = {f: frames[f] for f in plan.frames}

            def run_steps(n):
                keys = runner.exchange_keys([mine] * n)
                for i in range(n):
                    runner.run_sequence(mine, keys[i])
        else:
            def run_steps(n):
                for _ in range(n):
                    run_sequence(model, lattice, frames)

        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed = D.max_over_ranks(time.perf_counter() - t0, device=None if via_host else "cuda")

        # vertex counts of the workload (data dependent; printed with every result)
        model.reset_sequence()
        lat = make_lattice(contents)
        vcounts = []
        for t, (pos, val) in enumerate(frames):
            model(lat, pos, val, t != len(frames) - 1, False)
            l1 = lat.coarsen()
            vcounts.append([lat.nr_lattice_vertices(), l1.nr_lattice_vertices(), l1.coarsen().nr_lattice_vertices()])
        model.reset_sequence()

        # ---- roofline pass: per-launch HIP-event timing of the dominant kernel over the same workload ----
        roof = None
        if rank == 0:
            # (through the operator-level route, whose wrappers carry the HIP-event hooks; the frame program used in
            # the timed region launches the very same kernels with the same arguments, tests/test_gpu_engine.py)
            model.use_frame_program = False
            ops.profile_begin()
            reps = max(2, min(args.steps, 5))
            for _ in range(reps):
                run_sequence(model, lattice, frames)
            rec = ops.profile_end()
            model.use_frame_program = True
            breakdown = {}
            for name, ms, meta in rec:
                d = breakdown.setdefault(name, [0, 0.0])
                d[0] += 1
                d[1] += ms
            g = [(ms, meta) for name, ms, meta in rec if name == "gather_gemm"]
            tot_ms = sum(ms for ms, _ in g)
            tot_fl = sum(gemm_flops(m) for _, m in g)
            tot_by = sum(gemm_bytes(m) for _, m in g)
            big = [(ms, m) for ms, m in g if m["taps"] == 9 and m["cin"] == 192]
            big_tf = sum(gemm_flops(m) for _, m in big) / (sum(ms for ms, _ in big) * 1e-3) / 1e12 if big else None
            achieved = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
            roof = {"kernel": "k_gather_gemm", "bound": "mfma", "achieved": round(achieved, 3),
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": None, "launches_per_step": len(g) // reps,
                    "avg_launch_us": round(tot_ms * 1e3 / max(len(g), 1), 2),
                    "flops_per_launch": tot_fl / max(len(g), 1), "algorithmic_bytes_per_launch": tot_by / max(len(g), 1),
                    "largest_shape_TFLOPs": round(big_tf, 2) if big_tf else None,
                    "share_of_step_time": round((tot_ms / reps) / (elapsed / args.steps * 1e3), 3)}
            # HBM-side traffic per launch cannot be read from inside the process: it comes from the two rocprofv3 --pmc
            # passes of this same workload (FETCH_SIZE, WRITE_SIZE) summarised by tools/pmc_summary.py
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            default_workload = (args.points, args.frames, args.sigma, args.rnn) == (120000, 4, 0.6, "gru,gru,aflow,gru")
            if os.path.exists(pmc) and default_workload:
                with open(pmc) as f:
                    roof["traffic"] = round(json.load(f)["hbm_bytes_per_launch"])
                roof["traffic_unit"] = "B per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc passes, profiles/pmc_traffic.json)"
            if args.breakdown:
                for k, (cnt, ms) in sorted(breakdown.items(), key=lambda kv: -kv[1][1]):
                    print("  %-16s %5d calls/step %9.3f ms/step" % (k, cnt // reps, ms / reps), file=sys.stderr)
                shapes = {}
                for ms, m in g:
                    key = (m["M"] // 100 * 100, m["cin"], m["taps"], m["N"])
                    d = shapes.setdefault(key, [0, 0.0, 0.0])
                    d[0] += 1
                    d[1] += ms
                    d[2] += gemm_flops(m)
                print("  gather_gemm by shape (M~, cin, taps, N): calls/step, us/call, TFLOP/s", file=sys.stderr)
                for k, (cnt, ms, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:24]:
                    print("    %-22s %4.1f %8.1f %7.2f" % (k, cnt / reps, ms * 1e3 / cnt, fl / ms / 1e9), file=sys.stderr)

    groups = plan.nr_groups if frames_mode else args.gpus
    clouds = groups * args.steps * args.frames
    value = clouds / elapsed

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, contents, args)

    if rank == 0:
        par = ("frames of a sequence sharded over %d ranks (key all-gather + hidden-state hand-off), %d group(s)"
               % (plan.group_size, plan.nr_groups)) if frames_mode else \
            "one sequence stream per GPU (no data-path collective)"
        line = {
            "metric": "point-clouds/sec (120k pts, sigma=0.6, 4-frame seq)",
            "value": round(value, 3), "unit": "clouds/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d-frame sequence, %d pts/frame, sigma=%s, rnn_modules=[%s], 26 classes, "
                                   "full U-Net lattice encoder/decoder, inference" % (args.frames, args.points, args.sigma, args.rnn),
                       "parallelism": par, "vertices_per_frame_V0_V1_V2": vcounts},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(model, contents, args):
    """Times the CPU oracle (PyTorch eager restatement of the same path, same weights) on a bounded sample:
    whole sequences of the bench workload until at least --cpu-seconds of CPU work were done."""
    import torch
    from temporal_latticenet_amd.synthetic import make_sequence
    from oracle.model import OracleLNN
    m = contents["model"]
    cores = min(os.cpu_count() or 1, 32)     # eager ops on a few-thousand-row lattice do not scale past this
    torch.set_num_threads(cores)
    oracle = OracleLNN(model.state_dict(), 26, m["rnn_modules"], m["sequence_learning"], m["pointnet_layers"],
                       m["nr_downsamples"], m["nr_blocks_down_stage"], m["nr_blocks_bottleneck"],
                       m["nr_blocks_up_stage"], [args.sigma] * 3, 1 << 18, m["experiment"])
    seq = make_sequence(args.cpu_points, args.frames, seed=1234)
    done, t0 = 0, time.perf_counter()
    while True:
        oracle.reset_sequence()
        for t, (pos, val) in enumerate(seq):
            oracle.forward(pos, val, early_return=(t != len(seq) - 1))
        done += len(seq)
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or done >= 64:
            break
    return {"value": round(done / dt, 4), "unit": "clouds/s", "cores": cores, "kind": "port",
            "sample": "%d frames (%d whole %d-frame sequences of %d points, same config and weights), oracle/model.py, %.1f s"
                      % (done, done // len(seq), len(seq), args.cpu_points, dt)}


if __name__ == "__main__":
    main()
