#!/bin/bash
# the pool tests, then K2 alone under rocprofv3 (per-kernel averages): bash tools/pool_time.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "pool or pointnet" 2>&1 | tail -2
rm -rf gpurun_out/pp1 && mkdir -p gpurun_out/pp1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp1 -- python3 tools/pool_probe.py 5 2>/dev/null | tail -1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pp1/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "pool" in n: print("  %-40s calls %4s avg %8.1f us min %7.1f" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
rm -rf gpurun_out/pp1
