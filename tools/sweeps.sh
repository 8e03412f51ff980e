#!/bin/bash
# the configs table and the lattice-size sweep of DESIGN.md section 6 (run on the GPU box from the repo root)
set -e
mkdir -p gpurun_out/sweeps
: > gpurun_out/sweeps/configs.txt
for c in 1 2 3 headline 5 5b; do python tools/configs_bench.py $c >> gpurun_out/sweeps/configs.txt 2>&1; [ $c != 1 ] && python tools/configs_bench.py $c pairs >> gpurun_out/sweeps/configs.txt 2>&1; echo "config $c done"; done
: > gpurun_out/sweeps/sigma.txt
for s in 0.6 0.4 0.2 0.1 0.05; do for n in 1 4; do
  steps=60; [ "$s" = "0.1" ] && steps=20; [ "$s" = "0.05" ] && steps=8
  python bench.py --no-cpu-baseline --sigma $s --streams $n --pairs 0 --steps $steps --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('sigma $s streams $n clouds/s', d['value'], 'V', d['config']['vertices_per_frame_V0_V1_V2'][-1], 'gemm: %s TFLOP/s (%s of peak), %s us/launch' % (r['achieved'], r['frac'], r['avg_launch_us']))" >> gpurun_out/sweeps/sigma.txt
  echo "sigma $s x$n done"
done; done
