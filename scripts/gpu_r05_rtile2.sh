#!/bin/bash
# Round 5: the sorted row tiles, second form (old values of a tile's rows requested at once, descriptors / words ahead) -- headline sweep + kernel times
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_rtile2; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4), (d.get('stats') or {}).get('fused_epilogue'))"; }
echo "== parity"; timeout 600 python -m pytest tests/test_vertex_order.py -m gpu -x -q -k "sorted_row" 2>&1 | tail -2
H="python bench.py --no-cpu-baseline --no-extra"
for rows in 8192; do for e in 8192 16384 24576 32768 49152; do
  GRB_RTILE_ROWS=$rows GRB_RTILE_ENTRIES=$e timeout 300 $H > "$OUT/head_r${rows}_e$e.json" 2>"$OUT/head_r${rows}_e$e.err"; line "$OUT/head_r${rows}_e$e.json" "rows=$rows entries=$e"
done; done
timeout 300 python bench.py --workload mxv_min_plus --no-cpu-baseline --no-extra > "$OUT/unmasked.json" 2>/dev/null; line "$OUT/unmasked.json" "unmasked default"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rt -o h -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra > /dev/null 2>&1 ); cp $(find /tmp/prof_rt -name '*kernel_stats.csv' | head -1) "$OUT/headline_kernel_stats.csv"; grep -E "grb::k_mxv|grb::k_long_init|k_rows_tail" "$OUT/headline_kernel_stats.csv" | cut -c1-150 | head -8
