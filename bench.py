#!/usr/bin/env python3
"""bench.py — point-clouds/sec through the temporal-LatticeNet hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One STEP = one 4-frame sequence (120k points/frame, sigma 0.6, 26 classes, rnn_modules=[gru,gru,aflow,gru], the
reference's pretrained configuration) through LNN_SEQ.forward: distribute -> PointNet pool -> U-Net of lattice
convolutions with GRU/AFlow fusion -> slice, inference mode, inputs already resident in HBM — on EVERY one of the
--streams (default 4) independent sequence streams of a GPU (sequences are independent units, train_ln.py:236-239;
temporal_latticenet_amd/streams.py).  value = clouds of all streams and ranks / wall time.

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

# concurrent sequence streams want one hardware queue each (HIP's default of 4 is shared with the null stream, so two
# of four streams would serialise); read by the HIP runtime when it initialises, i.e. before the first device call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=120000)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--rnn", type=str, default="gru,gru,aflow,gru")
    ap.add_argument("--mode", choices=["sequences", "frames"], default="sequences")
    ap.add_argument("--streams", type=int, default=4,
                    help="independent sequences in flight per GPU (one HIP stream + host thread + model replica each); "
                         "one step = one sequence on every stream")
    ap.add_argument("--pairs", type=int, default=8,
                    help="1: every stream steps two sequences in lock-step with shared gather-GEMM launches (2..8: that "
                         "many); one step = that many sequences on every stream")
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
            # every stream gets its own synthetic drive (different seed => different vertex counts per stream)
            from temporal_latticenet_amd.streams import SequenceStreams
            S = max(1, args.streams)
            per = 1 if args.pairs <= 0 else (2 if args.pairs == 1 else min(args.pairs, 8))
            with quiet:
                pool = SequenceStreams(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents),
                                       frames, S, pairs=per if per > 1 else False)
            # one ray-cast drive per stream (a different seed each: different vertex counts); the further sequences of a
            # stream's lock-step group are that drive turned about the vertical axis (again other vertex counts, without
            # paying the CPU ray casting 32 times)
            drives = [frames] + [
                [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda())
                 for p, v in make_sequence(args.points, args.frames, seed=seed + 1000 * i)] for i in range(1, S)]

            def turned(drive, j):
                if j == 0:
                    return drive
                import math
                c, s_ = math.cos(0.7 * j), math.sin(0.7 * j)
                rot = torch.tensor([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]], device="cuda")
                return [((p @ rot.T).contiguous(), v) for p, v in drive]

            per_stream = [turned(drives[i], j) for i in range(S) for j in range(per)]

            def run_steps(n):
                pool.run([per_stream[per * i:per * i + per] * n for i in range(S)])

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

        # ---- roofline pass: the dominant kernel (gather-GEMM, fp32 MFMA) over the same workload.  The frame program
        # remembers the resolved arguments of every gather-GEMM launch of a frame; right after the frame those launches
        # are replayed back to back between two HIP events on the launch stream (tln_program_replay_gemms), so the
        # average covers kernel time plus the launch gap and nothing else.
        roof = None
        if rank == 0:
            reps = 5
            tot_ms, tot_n, tot_fl, tot_by = 0.0, 0, 0.0, 0.0
            lat = make_lattice(contents)
            for t, (pos, val) in enumerate(frames):
                model(lat, pos, val, t != len(frames) - 1, False)
                prog = getattr(model, "_program", None)
                if prog is None or not getattr(model, "_program_active", False):
                    break
                prog.capture_gemms(True)
                if t == 0:      # capture starts with the NEXT frame: run frame 0 again on a fresh lattice
                    model.reset_sequence()
                    lat = make_lattice(contents)
                    model(lat, pos, val, t != len(frames) - 1, False)
                ms, n, fl, by = prog.replay_gemms(reps)
                tot_ms, tot_n, tot_fl, tot_by = tot_ms + ms, tot_n + n, tot_fl + fl, tot_by + by
            model.reset_sequence()
            if getattr(model, "_program", None) is not None:
                model._program.capture_gemms(False)
            if tot_n:
                achieved = tot_fl / (tot_ms * 1e-3) / 1e12
                roof = {"kernel": "k_gather_gemm", "bound": "mfma", "achieved": round(achieved, 3),
                        "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": None, "launches_per_step": tot_n // reps,
                        "avg_launch_us": round(tot_ms * 1e3 / tot_n, 2),
                        "flops_per_launch": tot_fl / tot_n, "algorithmic_bytes_per_launch": tot_by / tot_n,
                        "note": "every gather-GEMM op of one 4-frame sequence (the GRU cell's two internal products "
                                "excluded), replayed back to back on one stream running alone"}
                # HBM-side traffic per launch cannot be read from inside the process: it comes from the two rocprofv3
                # --pmc passes of this same workload (FETCH_SIZE, WRITE_SIZE) summarised by tools/pmc_summary.py
                pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                default_workload = (args.points, args.frames, args.sigma, args.rnn) == (120000, 4, 0.6, "gru,gru,aflow,gru")
                if os.path.exists(pmc) and default_workload:
                    with open(pmc) as f:
                        roof["traffic"] = round(json.load(f)["hbm_bytes_per_launch"])
                    roof["traffic_unit"] = "B per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc passes, " \
                                           "profiles/pmc_traffic.json)"
            if args.breakdown:
                model.use_frame_program = False
                ops.profile_begin()
                for _ in range(3):
                    run_sequence(model, lattice, frames)
                rec = ops.profile_end()
                model.use_frame_program = True
                breakdown, shapes = {}, {}
                for name, ms, meta in rec:
                    d = breakdown.setdefault(name, [0, 0.0])
                    d[0] += 1
                    d[1] += ms
                    if name == "gather_gemm":
                        key = (meta["M"] // 100 * 100, meta["cin"], meta["taps"], meta["N"])
                        e = shapes.setdefault(key, [0, 0.0, 0.0])
                        e[0] += 1
                        e[1] += ms
                        e[2] += gemm_flops(meta)
                for k, (cnt, ms) in sorted(breakdown.items(), key=lambda kv: -kv[1][1]):
                    print("  %-16s %5d calls/step %9.3f ms/step" % (k, cnt // 3, ms / 3), file=sys.stderr)
                print("  gather_gemm by shape (M~, cin, taps, N): calls/step, us/call (HIP events around each operator "
                      "call), TFLOP/s", file=sys.stderr)
                for k, (cnt, ms, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:24]:
                    print("    %-22s %4.1f %8.1f %7.2f" % (k, cnt / 3, ms * 1e3 / cnt, fl / ms / 1e9), file=sys.stderr)

    per_stream_seqs = 1 if args.pairs <= 0 else (2 if args.pairs == 1 else min(args.pairs, 8))
    groups = plan.nr_groups if frames_mode else args.gpus * max(1, args.streams) * per_stream_seqs
    clouds = groups * args.steps * args.frames
    value = clouds / elapsed

    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, contents, args)

    if rank == 0:
        par = ("frames of a sequence sharded over %d ranks (key all-gather + hidden-state hand-off), %d group(s)"
               % (plan.group_size, plan.nr_groups)) if frames_mode else \
            "%d independent sequence stream(s) per GPU, each on its own HIP stream; one step = %s %d-frame sequence%s on " \
            "every stream (no data-path collective)" % (max(1, args.streams), ("%d lock-stepped" % per_stream_seqs) if args.pairs else "one",
                                                        args.frames, "s" if args.pairs else "")
        line = {
            "metric": "point-clouds/sec (120k pts, sigma=0.6, 4-frame seq)",
            "value": round(value, 3), "unit": "clouds/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if frames_mode or per_stream_seqs == 1 else
                    "synthetic (one ray-cast drive per stream; the further sequences of its lock-step group are that "
                    "drive turned about the vertical axis)",
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
    oracle.exact_pool = False                # the timed baseline is plain PyTorch-CPU eager (F.linear)
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
