#!/bin/bash
# The per-round profile set (run on the GPU box from the repo root): bench line, rocprofv3 kernel stats with 4 and 1
# sequence streams, HBM-side traffic from two separate --pmc passes.  usage: bash tools/profile_set.sh <tag> [pmc-only]
set -e
TAG=${1:-r01_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" != "pmc-only" ]; then
  mkdir -p $O/ks4 $O/ks1 $O/ks1g
  python bench.py > $O/bench.json 2> $O/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks4 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof_4streams.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --pairs 0 > $O/bench_under_rocprof_1stream.json 2>/dev/null
  # one stream stepping a lock-step group of 8 alone: the launches the `roofline` headline times, without other streams
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1g -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --pairs 8 > $O/bench_under_rocprof_1stream_group8.json 2>/dev/null
  cp $(ls $O/ks4/*/*kernel_stats.csv | head -1) $O/kernel_stats_4streams.csv
  cp $(ls $O/ks1g/*/*kernel_stats.csv | head -1) $O/kernel_stats_1stream_group8.csv
  cp $(ls $O/ks1/*/*kernel_stats.csv | head -1) $O/kernel_stats_1stream.csv
  rm -rf $O/ks4 $O/ks1 $O/ks1g
  cat $O/bench.json
fi
mkdir -p $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --pairs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --pairs 0 > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc $O/pmc_traffic.csv $O/pmc_traffic.json
rm -rf $O/pmc
# the same two passes in the mode the timed region launches the stages in: one stream stepping a lock-step group of eight
# (every K1 / K2 / K8 launch carries eight frames) -> bytes per frame of the batched launches
mkdir -p $O/pmcg/FETCH_SIZE $O/pmcg/WRITE_SIZE
# (tools/group_run.py: ONLY batched launches — bench.py mixes in its one-sequence replays and checks)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcg/FETCH_SIZE -- python3 tools/group_run.py 6 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcg/WRITE_SIZE -- python3 tools/group_run.py 6 8 > /dev/null 2>&1
python tools/pmc_summary.py $O/pmcg $O/pmc_traffic_group.csv $O/pmc_traffic_group.json 8
rm -rf $O/pmcg
head -5 $O/pmc_traffic.csv
