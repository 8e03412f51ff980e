#!/bin/bash
# throughput-mode sweep of the tiled-kernel hand-over (TLN_GEMM_HEAVY="mode,min_tiles,min_chunks"), 4 streams
for cfg in "$@"; do
  v=$(TLN_GEMM_HEAVY=$cfg timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -o "\"value\": [0-9.]*") || exit 1
  echo "$cfg $v"
done
