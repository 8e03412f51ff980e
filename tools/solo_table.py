#!/usr/bin/env python3
"""Per-frame kernel table from a rocprofv3 --kernel-trace csv of tools/solo_run.py: the last `frac` of the trace, kernels
summed per name, divided by the number of frames in that part (counted by k_slice_deform launches).
  python tools/solo_table.py DIR_OR_CSV [frac=0.6]"""
import collections
import csv
import glob
import os
import re
import sys

src = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t1 - int((t1 - t0) * frac)
ev = [e for e in ev if e[0] >= lo]
# whole frames only: from the first launch after a slice head to the last slice head
heads = [i for i, e in enumerate(ev) if "k_slice_deform" in e[2]]
ev = ev[heads[0] + 1:heads[-1] + 1]
frames = len(heads) - 1
wall = ev[-1][1] - ev[0][0]
tot = collections.defaultdict(lambda: [0, 0])
busy = 0
last_end = ev[0][0]
for s, e, n in ev:
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((GemmArgs|[A-Za-z]+Jobs|GnFin).*", "", n)[:64]
    tot[n][0] += e - s
    tot[n][1] += 1
    if e > last_end:
        busy += e - max(s, last_end)
        last_end = e
ksum = sum(v[0] for v in tot.values())
print("%d frames: wall %.1f us per frame, some kernel running %.1f us (%.1f %%), kernel time summed %.1f us, %.1f launches per frame"
      % (frames, wall / frames / 1e3, busy / frames / 1e3, 100.0 * busy / wall, ksum / frames / 1e3, len(ev) / frames))
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("  %-64s %6.1f us/frame  %5.1f launches/frame  %6.1f us each  %4.1f %%" % (n, d / frames / 1e3, c / frames, d / c / 1e3, 100.0 * d / ksum))
