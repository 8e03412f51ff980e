#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the PMC slots require) into
HBM-side bytes per launch per kernel.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-B
requests at 64 B for wide (16 B/lane) reads, so fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact.
usage: pmc_summary.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <out.csv> [<out.json> [frames per launch]]
frames per launch: 8 when the passes ran a lock-step group of eight (bench.py --streams 1 --pairs 8): every K1 / K2 / K8
launch then carries the frames of eight sequences (blockIdx.y = sequence) and the per-frame figures divide by it."""
import csv
import glob
import json
import re
import sys


def load(d, counter, full_only=False):
    """per kernel [launches, sum of the counter].  full_only (group mode): launches whose grid is less than a quarter of
    the kernel's largest are left out — the stream pool warms its model replicas on 4096-point slices, one frame per
    launch, and those must not count as batched launches of eight frames"""
    f = glob.glob("%s/%s/*/*counter_collection.csv" % (d, counter))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    gmax = {}
    for r in rows:
        k = name(r)
        gmax[k] = max(gmax.get(k, 0), int(r["Grid_Size"]))
    acc = {}
    for r in rows:
        k = name(r)
        if full_only and int(r["Grid_Size"]) * 4 < gmax[k]:
            continue
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    d, out = sys.argv[1], sys.argv[2]
    group = len(sys.argv) > 4 and int(sys.argv[4]) > 1
    fe, wr = load(d, "FETCH_SIZE", group), load(d, "WRITE_SIZE", group)
    rows = []
    for k in sorted(set(fe) | set(wr)):
        n = fe.get(k, wr.get(k))[0]
        fb = 2.0 * 1024 * fe.get(k, [1, 0.0])[1] / max(fe.get(k, [1])[0], 1)
        wb = 1024.0 * wr.get(k, [1, 0.0])[1] / max(wr.get(k, [1])[0], 1)
        rows.append((k, n, fb, wb, fb + wb))
    rows.sort(key=lambda r: -r[4] * r[1])
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "fetch_bytes_per_launch(2x FETCH_SIZE KB)", "write_bytes_per_launch", "hbm_bytes_per_launch"])
        for r in rows:
            w.writerow([r[0], r[1]] + ["%.0f" % x for x in r[2:]])
    g = [r for r in rows if r[0].startswith("k_gather_gemm")]
    n = sum(r[1] for r in g)
    total = sum(r[4] * r[1] for r in g)
    summary = {"kernel": "k_gather_gemm", "launches": n, "hbm_bytes_per_launch": total / n if n else None,
               "correction": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE halves wide reads)"}
    # the scatter / gather stages: HBM-side bytes per FRAME (sum over the stage's kernels / frames profiled).  K1 =
    # the partitioned kernels (k_bk_*) or, with TLN_K1_LEGACY=1, the per-row-atomic ones.
    stages = {"K1_distribute": ("k_bk_split", "k_bk_insert", "k_bk_prefix", "k_bk_place", "k_distribute_insert",
                                "k_bins_alloc", "k_bins_scatter", "k_bins_mean"),
              "K2_pointnet_pool": ("k_pool_bins", "k_pool_bins_finalize", "k_pool_chunks", "k_pool_finalize"),
              "K8_slice": ("k_slice_deform", "k_slice_gather", "k_slice")}
    fpl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    frames = fpl * sum(r[1] for r in rows if r[0].startswith("k_distribute_insert") or r[0].startswith("k_bk_split"))
    last = fpl * sum(r[1] for r in rows if r[0].startswith("k_slice_deform"))
    summary["frames_per_launch"] = fpl
    sc = {}
    for name, ks in stages.items():
        tot = sum(r[4] * r[1] for r in rows if any(r[0] == k or r[0].startswith(k + "<") for k in ks))
        den = last if name == "K8_slice" else frames
        if den:
            sc[name] = tot / den
    summary["scatter_hbm_bytes_per_frame"] = sc
    print(json.dumps(summary))
    if len(sys.argv) > 3:
        json.dump(summary, open(sys.argv[3], "w"))


if __name__ == "__main__":
    main()
