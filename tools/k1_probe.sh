#!/bin/bash
# per-kernel K1 durations for a list of environment variants: bash tools/k1_probe.sh "TLN_BK_PPB=512" "TLN_K1_LEGACY=1" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  rm -rf gpurun_out/k1p && mkdir -p gpurun_out/k1p
  echo "=== variant: ${v:-default}"
  for kv in $v; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1p -- python3 tools/k1_probe.py 2>/dev/null | tail -1
  for kv in $v; do unset "${kv%%=*}"; done
  python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/k1p/*/*kernel_stats.csv")[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if n.startswith("k_") or "rocclr" in n:
        print("  %-28s calls %4s avg %8.1f us  min %7.1f max %7.1f" % (n[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
rm -rf gpurun_out/k1p
