#!/usr/bin/env python3
"""Device memory per resident sequence, by owner (VERDICT r3 item 9 / "next" 7).

One 4 x 120k-point sequence through the frame program (capacity as bench.py: 1 << 18 unless given), then the timed
configuration's pool of 4 streams x 8 lock-stepped sequences: what the lattice level stack, the frame program and PyTorch's
caching allocator hold, against the HIP runtime's own used-memory figure.
  python tools/mem_report.py [capacity=262144|auto] [streams=4] [per=8]"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from temporal_latticenet_amd.configs import build_model, make_config, make_lattice  # noqa: E402
from temporal_latticenet_amd.streams import SequenceStreams  # noqa: E402
from temporal_latticenet_amd.workload import group_sequences, stream_drives  # noqa: E402

cap = sys.argv[1] if len(sys.argv) > 1 else str(1 << 18)
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cap = "auto" if cap == "auto" else int(cap)
MB = 1.0 / (1 << 20)


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


contents = make_config(capacity=cap)
quiet = contextlib.redirect_stdout(io.StringIO())
mk_lat = (lambda: make_lattice(contents, nr_points=120000, frames=4)) if cap == "auto" else (lambda: make_lattice(contents))
base = used()
with quiet:
    model = build_model(contents).eval()
drives = stream_drives(120000, 4, 1234, S)
first = drives[0]
with torch.no_grad():
    lat = mk_lat()
    for t, (p, v) in enumerate(first[:2]):
        model(lat, p[:4096], v[:4096], t != 1, False)
    model.reset_sequence()
    after_model = used()
    lat = mk_lat()
    for t, (p, v) in enumerate(first):
        a, b, lat = model(lat, p, v, t != 3, False)
    one = used()
rep = {"capacity": lat.capacity(), "V0": lat.nr_lattice_vertices(),
       "lattice_MB": {k: round(v * MB, 1) for k, v in lat.memory_bytes().items()},
       "program_MB": {k: round(v * MB, 1) for k, v in model._program.memory_bytes().items()},
       "inputs_and_model_MB": round((after_model - base) * MB, 1),
       "one_sequence_device_delta_MB": round((one - after_model) * MB, 1),
       "torch_allocated_MB": round(torch.cuda.memory_allocated() * MB, 1),
       "torch_reserved_MB": round(torch.cuda.memory_reserved() * MB, 1)}
print(json.dumps(rep))
model.reset_sequence()
del lat, a, b
before_pool = used()
with quiet:
    pool = SequenceStreams(model, lambda: build_model(contents).eval(), mk_lat, first, S, pairs=per)
seqs = group_sequences(drives, per)
with torch.no_grad():
    pool.run([seqs[per * i:per * i + per] for i in range(S)])
    pool.run([seqs[per * i:per * i + per] for i in range(S)])
after_pool = used()
lat_tot = sum(l.memory_bytes()["total"] for l in pool.lattices)
prog_tot = sum(m._program.memory_bytes()["total"] for m in pool.models if getattr(m, "_program", None) is not None)
rep2 = {"resident_sequences": S * per, "device_delta_MB": round((after_pool - before_pool) * MB, 1),
        "per_sequence_MB": round((after_pool - before_pool) * MB / (S * per), 1),
        "lattices_MB": round(lat_tot * MB, 1), "programs_MB": round(prog_tot * MB, 1),
        "torch_allocated_MB": round(torch.cuda.memory_allocated() * MB, 1),
        "torch_reserved_MB": round(torch.cuda.memory_reserved() * MB, 1),
        "device_used_total_MB": round(after_pool * MB, 1)}
print(json.dumps(rep2))
pool.close()
