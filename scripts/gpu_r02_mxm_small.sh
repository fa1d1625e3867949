#!/bin/bash
# small SpGEMM sizes: unit kernels (1) against the round-1 row kernels (0)
for sc in 12 14 16; do for wl in mxm_plus_times mxm_plus_times_masked; do for hk in 1 0; do
  GRB_MXM_HEAVY_KERNEL=$hk python bench.py --workload $wl --scale $sc --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl scale $sc heavy_kernel=$hk', round(d['ms_per_step'],3))"
done; done; done
