#!/bin/bash
# Round 5, late: do the round-4/5 layouts (vertex order, hub level, dictionary, row tiles) pay below 48 Mi entries?  scale 20 / 21 / 22 with the size thresholds lowered
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_small; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4), {k: (d.get('stats') or {}).get(k) for k in ('ordered','fused_epilogue','long_kernel','value_dict')})"; }
for s in 20 21 22; do for wl in mxv_min_plus mxv_min_plus_masked; do
  timeout 300 python bench.py --scale $s --workload $wl --no-cpu-baseline --no-extra > "$OUT/s${s}_${wl}_default.json" 2>/dev/null; line "$OUT/s${s}_${wl}_default.json" "s$s $wl default"
  GRB_ORDER_MIN_NNZ=1000000 GRB_LEAN_MIN_NNZ=1000000 timeout 300 python bench.py --scale $s --workload $wl --no-cpu-baseline --no-extra > "$OUT/s${s}_${wl}_lowered.json" 2>/dev/null; line "$OUT/s${s}_${wl}_lowered.json" "s$s $wl order+lean from 1 M entries"
done; done
