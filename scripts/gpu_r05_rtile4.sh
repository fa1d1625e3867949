#!/bin/bash
# Round 5: k_mxv_rtile build-time ablations (results of the ablated builds are WRONG on purpose: verified is expected to be False) and workgroup sizes
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_rtile4; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4), (d.get('stats') or {}).get('fused_epilogue'))"; }
H="python bench.py --no-cpu-baseline --no-extra"
for v in rt_d1_w6 rt_abl1 rt_abl2 rt_abl4 rt_abl3 rt_abl7 rt_blk256 rt_blk1024; do
  GRB_MI355X_LIB=build/variants/$v/libgrb_mi355x.so timeout 300 $H > "$OUT/head_${v}.json" 2>"$OUT/head_${v}.err"; line "$OUT/head_${v}.json" "$v masked"
  GRB_MI355X_LIB=build/variants/$v/libgrb_mi355x.so timeout 300 $H --workload mxv_min_plus > "$OUT/unm_${v}.json" 2>"$OUT/unm_${v}.err"; line "$OUT/unm_${v}.json" "$v unmasked"
done
