#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME -> DIR/**/NAME_results.db).
  python tools/prof_stats.py gpurun_out/r2c_prof [top_n]"""
import glob
import os
import re
import sqlite3
import sys

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dbs = [root] if root.endswith(".db") else glob.glob(os.path.join(root, "**", "*.db"), recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from %s d join %s s "
                        "on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks)))
tot = sum(r[2] for r in rows)
print("kernel,calls,total_ms,avg_us,percent   (all kernels: %.3f ms)" % (tot / 1e6))
for n, c, t, a in rows[:top]:
    n = re.sub(r"\(.*", "", n).replace(".kd", "")
    print("%s,%d,%.3f,%.1f,%.1f" % (n[:100], c, t / 1e6, a / 1e3, 100 * t / tot))
