#!/usr/bin/env python3
"""How busy is the GPU during a traced run: from a rocprofv3 --kernel-trace csv, the fraction of the traced interval with
at least one kernel running, the mean number of kernels in flight, and the idle gaps.
  python tools/trace_coverage.py DIR_OR_CSV [skip_fraction_at_start [end_fraction]]
(bench.py: the timed region lies between its warm-up and its measurement replays — e.g. 0.3 0.5 of a --steps 40 run)"""
import csv
import glob
import os
import sys

src = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
stop = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + int((t1 - t0) * skip)          # the steady part (warm-up and setup skipped)
hi = t0 + int((t1 - t0) * stop)
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, _ in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
busy = 0
depth = 0
area = 0
last = t0
gaps = []
for t, d in pts:
    if depth > 0:
        busy += t - last
        area += depth * (t - last)
    elif t > last:
        gaps.append(t - last)
    depth += d
    last = t
wall = t1 - t0
print("traced %.2f ms, %d kernels; busy %.1f %%, mean kernels in flight %.2f (while busy %.2f)" % (
    wall / 1e6, len(ev), 100.0 * busy / wall, area / wall, area / max(busy, 1)))
gaps.sort(reverse=True)
print("idle gaps: %d, total %.2f ms, largest %s us" % (len(gaps), sum(gaps) / 1e6, [round(g / 1e3, 1) for g in gaps[:8]]))
