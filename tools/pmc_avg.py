#!/usr/bin/env python3
"""average PMC counter values per kernel from rocprofv3 --pmc csv output: pmc_avg.py <dir> [kernel substring]"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        a = acc[(r["Kernel_Name"][:50], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-50s %-34s %14.1f  (n=%d)" % (k, c, s / n, n))
