#!/bin/bash
# Round 5: k_mxv_rtile_bool on the BFS level step (configs[2]) and the BFS loop; parity on the GPU
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_bool; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4), (d.get('stats') or {}).get('fused_epilogue'), (d.get('config') or {}).get('cold_ms'))"; }
echo "== parity"; timeout 900 python -m pytest tests/test_vertex_order.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -2
H="python bench.py --workload mxv_lor_land_masked --no-cpu-baseline --no-extra"
GRB_ROWS_TILE=0 timeout 300 $H > "$OUT/step_tile0.json" 2>/dev/null; line "$OUT/step_tile0.json" "BFS step rows_tile=0"
for rows in 8192 16384; do for e in 32768 65536; do
  GRB_RTILE_ROWS=$rows GRB_RTILE_ENTRIES=$e timeout 300 $H > "$OUT/step_r${rows}_e$e.json" 2>/dev/null; line "$OUT/step_r${rows}_e$e.json" "BFS step rows=$rows entries=$e"
done; done
GRB_ROWS_TILE=0 timeout 600 python bench.py --workload bfs --steps 5 --no-cpu-baseline > "$OUT/bfs_tile0.json" 2>/dev/null; line "$OUT/bfs_tile0.json" "bfs loop rows_tile=0"
timeout 600 python bench.py --workload bfs --steps 5 --no-cpu-baseline > "$OUT/bfs.json" 2>/dev/null; line "$OUT/bfs.json" "bfs loop"
timeout 600 python bench.py --workload sssp --steps 3 --no-cpu-baseline > "$OUT/sssp.json" 2>/dev/null; line "$OUT/sssp.json" "sssp loop"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o h -- python $OLDPWD/bench.py --workload mxv_lor_land_masked --no-cpu-baseline --no-extra > /dev/null 2>&1 ); cp $(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) "$OUT/lor_land_kernel_stats.csv"; grep -E "grb::k_mxv|grb::k_long|k_rows_tail|k_pack" "$OUT/lor_land_kernel_stats.csv" | cut -c1-140 | head -8
