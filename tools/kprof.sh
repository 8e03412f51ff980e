#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) and HBM-side bytes per launch (two separate --pmc passes:
# FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE) of one python tool.  Run on the GPU box from the repo root:
#   bash tools/kprof.sh <out-prefix under gpurun_out/> <filter regex on kernel names> tools/k1_bench.py 8 5
set -e
OUT=$1; FILT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/$OUT.tmp
rm -rf $D && mkdir -p $D/ks $D/pmc/FETCH_SIZE $D/pmc/WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -- python3 "$@" > $D/run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc/FETCH_SIZE -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc/WRITE_SIZE -- python3 "$@" > /dev/null 2>&1
python3 tools/pmc_summary.py $D/pmc $D/traffic.csv > /dev/null
python3 - "$D" "$FILT" > gpurun_out/$OUT.txt <<'PY'
import csv, glob, re, sys
d, filt = sys.argv[1], re.compile(sys.argv[2])
tr = {}
for r in csv.DictReader(open(d + "/traffic.csv")):
    tr[r["kernel"]] = (float(r[list(r.keys())[2]]), float(r[list(r.keys())[3]]))
f = glob.glob(d + "/ks/*/*kernel_stats.csv")[0]
print("%-44s %6s %10s %10s %10s | %12s %12s" % ("kernel", "calls", "avg us", "min us", "max us", "fetch MB", "write MB"))
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    if not filt.search(n):
        continue
    fe, wr = tr.get(n, (float("nan"), float("nan")))
    print("%-44s %6s %10.1f %10.1f %10.1f | %12.2f %12.2f" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                      float(r["MaxNs"]) / 1e3, fe / 1e6, wr / 1e6))
PY
tail -3 $D/run.log >> gpurun_out/$OUT.txt
rm -rf $D
cat gpurun_out/$OUT.txt
