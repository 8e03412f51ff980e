#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 --pmc results database.
  python tools/pmc_kernels.py DIR [kernel-name substring]"""
import glob
import os
import re
import sqlite3
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    info = [t for t in tabs if "info_pmc" in t]
    if not pmc or not info:
        continue
    q = ("select s.kernel_name, i.name, count(*), avg(e.value) from %s e join %s d on e.event_id = d.id join %s s on "
         "d.kernel_id = s.id join %s i on e.pmc_id = i.id group by s.kernel_name, i.name" % (pmc[0], kd, ks, info[0]))
    try:
        rows = list(cur.execute(q))
    except sqlite3.OperationalError as e:
        print(db, "query failed:", e, tabs)
        continue
    by = {}
    for kn, cn, n, v in rows:
        if flt in kn:
            by.setdefault(re.sub(r"\(.*", "", kn)[:90], {})[cn] = (n, v)
    for kn, d in by.items():
        print(kn)
        for cn, (n, v) in sorted(d.items()):
            print("   %-34s %14.1f   (%d dispatches)" % (cn, v, n))
