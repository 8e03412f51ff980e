#!/bin/bash
# the fused GRU cell's launches in the one-sequence bench under rocprofv3, for the environment variants given:
#   bash tools/gru_time.sh "TLN_GRU_GRID_Y=2" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  rm -rf gpurun_out/gt && mkdir -p gpurun_out/gt
  echo "=== variant: ${v:-default}"
  for kv in $v; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --pairs 0 > /dev/null 2>&1
  for kv in $v; do unset "${kv%%=*}"; done
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gt/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "gru" in r["Name"]:
        print("  %-40s calls %4s avg %8.1f us  min %7.1f max %7.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
rm -rf gpurun_out/gt
