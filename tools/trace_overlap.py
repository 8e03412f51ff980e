#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: how busy the GPU was and how many kernels overlapped (time-weighted).
usage: trace_overlap.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = []
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
lo = t0 + (t1 - t0) * skip          # the timed region is the tail of the run
per = defaultdict(float)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= lo:
        continue
    s = max(s, lo)
    ev.append((s, 1))
    ev.append((e, -1))
    per[r["Kernel_Name"][:60]] += e - s
ev.sort()
hist = defaultdict(float)
cur, last = 0, lo
for t, d in ev:
    hist[cur] += t - last
    last = t
    cur += d
tot = t1 - lo
print("window %.1f ms" % (tot / 1e6))
for k in sorted(hist):
    print("  %2d kernels in flight: %5.1f%%" % (k, 100 * hist[k] / tot))
print("sum of kernel durations / window = %.2f" % (sum(per.values()) / tot))
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-60s %5.1f%% of window" % (k, 100 * v / tot))
