#!/bin/bash
# K1 in batched mode (eight frames per launch, tools/k1_bench.py 8 <reps> batched): per-kernel durations (rocprofv3
# --kernel-trace --stats) and HBM-side bytes (two --pmc passes: FETCH_SIZE, WRITE_SIZE; gfx950 corrections as in
# tools/pmc_summary.py) for a list of environment variants:
#   bash tools/k1_batched_probe.sh "TLN_K1_XCD=0" ...      (the first variant is always the default build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  rm -rf gpurun_out/k1b && mkdir -p gpurun_out/k1b
  echo "=== variant: ${v:-default}"
  for kv in $v; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1b/ks -- python3 tools/k1_bench.py 8 6 batched > gpurun_out/k1b/ks.log 2>&1 || { tail -5 gpurun_out/k1b/ks.log; exit 1; }
  grep -a "us per frame" gpurun_out/k1b/ks.log | tail -2
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/k1b/FETCH_SIZE -- python3 tools/k1_bench.py 8 3 batched > gpurun_out/k1b/f.log 2>&1 || { tail -5 gpurun_out/k1b/f.log; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/k1b/WRITE_SIZE -- python3 tools/k1_bench.py 8 3 batched > gpurun_out/k1b/w.log 2>&1 || { tail -5 gpurun_out/k1b/w.log; exit 1; }
  for kv in $v; do unset "${kv%%=*}"; done
  python3 - <<'PY'
import csv, glob, re
name = lambda s: re.sub(r"\(.*", "", s).replace("void ", "")
dur = {}
for r in csv.DictReader(open(glob.glob("gpurun_out/k1b/ks/*/*kernel_stats.csv")[0])):
    dur[name(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
def pmc(counter):
    acc = {}
    rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/k1b/%s/*/*counter_collection.csv" % counter)[0])) if r["Counter_Name"] == counter]
    gmax = {}
    for r in rows:
        gmax[name(r["Kernel_Name"])] = max(gmax.get(name(r["Kernel_Name"]), 0), int(r["Grid_Size"]))
    for r in rows:
        k = name(r["Kernel_Name"])
        if int(r["Grid_Size"]) * 4 < gmax[k]:
            continue
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}
fe, wr = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
print("%-20s %6s %9s %9s %9s | %12s %12s" % ("kernel", "calls", "avg us", "min us", "max us", "fetch MB", "write MB"))
tot = [0.0, 0.0, 0.0]
for k in ("k_bk_split", "k_bk_insert", "k_bk_prefix", "k_bk_place"):
    if k in dur:
        f, w = 2 * 1024 * fe.get(k, 0.0) / 1e6, 1024 * wr.get(k, 0.0) / 1e6
        print("%-20s %6d %9.1f %9.1f %9.1f | %12.2f %12.2f" % ((k,) + dur[k] + (f, w)))
        tot[0] += dur[k][1]; tot[1] += f; tot[2] += w
print("%-20s %6s %9.1f %19s | %12.2f %12.2f   per frame: %.1f us, %.1f MB" % ("K1 (8 frames)", "", tot[0], "", tot[1], tot[2], tot[0] / 8, (tot[1] + tot[2]) / 8))
PY
done
rm -rf gpurun_out/k1b
