#!/bin/bash
# Round 5: k_mxv_rtile -- wavefronts per SIMD and pipeline depth (build variants), tile sizes
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_rtile3; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4), (d.get('stats') or {}).get('fused_epilogue'))"; }
H="python bench.py --no-cpu-baseline --no-extra"
for v in shipped rt_d1_w6 rt_d2_w6 rt_d2_w5; do for e in 32768 65536; do
  lib=""; [ $v != shipped ] && lib="build/variants/$v/libgrb_mi355x.so"
  GRB_MI355X_LIB=$lib GRB_RTILE_ENTRIES=$e timeout 300 $H > "$OUT/head_${v}_e$e.json" 2>"$OUT/head_${v}_e$e.err"; line "$OUT/head_${v}_e$e.json" "$v entries=$e"
done; done
for v in shipped rt_d2_w6; do
  lib=""; [ $v != shipped ] && lib="$PWD/build/variants/$v/libgrb_mi355x.so"
  ( cd /tmp && GRB_MI355X_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o h -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra > /dev/null 2>&1 ); cp $(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats_$v.csv"; echo $v; grep -E "grb::k_mxv_rtile|grb::k_mxv_hstrip|grb::k_mxv_ctile" "$OUT/kernel_stats_$v.csv" | cut -c1-130
done
