#!/bin/bash
# BFS step (BOOL items kernel): long-row threshold sweep
for t in 128 192 256 384 512 1024; do
  GRB_SPLIT_MIN_LEN=$t python bench.py --workload mxv_lor_land_masked --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T=$t', round(d['ms_per_step'],4), d.get('verified'))"
done
