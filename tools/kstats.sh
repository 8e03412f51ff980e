#!/bin/bash
# rocprofv3 kernel stats of one bench.py configuration: bash tools/kstats.sh <out name under gpurun_out/> <bench args...>
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/$OUT.tmp
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$OUT.json 2>/dev/null
cp $(ls $D/*/*kernel_stats.csv | head -1) gpurun_out/$OUT.csv
rm -rf $D
python3 - gpurun_out/$OUT.csv <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
gemm = sum(float(r["TotalDurationNs"]) for r in rows if "gather_gemm" in r["Name"])
print("kernel time %.1f ms, gather-GEMM share %.1f %%, other %.1f %%" % (tot / 1e6, 100 * gemm / tot, 100 - 100 * gemm / tot))
for r in rows[:34]:
    print("%-70s %6s %9.1f us %6.2f%%" % (re.sub(r"\(.*", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
