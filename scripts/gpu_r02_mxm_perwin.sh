#!/bin/bash
# SpGEMM: products per window from which a row is walked as units (hash rows below)
mkdir -p gpurun_out/mxmwave
for sc in 20 22; do for wl in mxm_plus_times mxm_plus_times_masked; do for pw in 32 16 8 4; do
  GRB_MXM_UNIT_MIN_PER_WINDOW=$pw python bench.py --workload $wl --scale $sc --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl scale $sc per_window=$pw', round(d['ms_per_step'],2))"
done; done; done
