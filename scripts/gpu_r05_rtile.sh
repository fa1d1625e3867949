#!/bin/bash
# Round 5, second GPU call: the sorted row tiles of the short rows (k_mxv_rtile) -- parity on the GPU, the headline with / without them,
# tile heights and sizes, per-kernel times, and the honest-input lines of the default command.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_rtile; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified','layout_build_call_ms','preprocess_bytes')}, round((d.get('roofline') or {}).get('frac') or 0, 4), (d.get('stats') or {}).get('fused_epilogue'))"; }
echo "== pytest -m gpu (order / parity / scale)"; ( time timeout 900 python -m pytest tests/test_vertex_order.py tests/test_gpu_scale.py tests/test_gxb_ingress.py -m gpu -x -q ) > "$OUT/tests.log" 2>&1; tail -4 "$OUT/tests.log"
H="python bench.py --no-cpu-baseline --no-extra"
GRB_ROWS_TILE=0 timeout 300 $H > "$OUT/head_tile0.json" 2>"$OUT/head_tile0.err"; line "$OUT/head_tile0.json" "rows_tile=0"
for rows in 8192 16384; do for e in 16384 32768 65536; do
  GRB_RTILE_ROWS=$rows GRB_RTILE_ENTRIES=$e timeout 300 $H > "$OUT/head_r${rows}_e$e.json" 2>"$OUT/head_r${rows}_e$e.err"; line "$OUT/head_r${rows}_e$e.json" "rows=$rows entries=$e"
done; done
timeout 300 python bench.py --workload mxv_min_plus --no-cpu-baseline --no-extra > "$OUT/unmasked.json" 2>/dev/null; line "$OUT/unmasked.json" "unmasked default"
GRB_ROWS_TILE=0 timeout 300 python bench.py --workload mxv_min_plus --no-cpu-baseline --no-extra > "$OUT/unmasked_tile0.json" 2>/dev/null; line "$OUT/unmasked_tile0.json" "unmasked rows_tile=0"
echo "== rocprofv3 kernel stats of the default headline"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rt -o h -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra > /dev/null 2>&1 ); cp $(find /tmp/prof_rt -name '*kernel_stats.csv' | head -1) "$OUT/headline_kernel_stats.csv"; grep -E "grb::k_mxv|grb::k_long_init|k_rows_tail|k_rtile|k_tag|radix" "$OUT/headline_kernel_stats.csv" | cut -c1-170 | head -20
echo "== sssp loop (absorbing fill -> tiles)"; timeout 600 python bench.py --workload sssp --steps 3 --no-cpu-baseline > "$OUT/sssp.json" 2>/dev/null; line "$OUT/sssp.json" "sssp"
echo "== default command with extras"; ( time timeout 900 python bench.py --no-cpu-baseline ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['roofline']['frac'], d['verified'], d['config']['workload'][:200])
for e in d.get('extra',[]): print('  extra', e.get('workload','')[:90], e.get('ms_per_step'), (e.get('roofline') or {}).get('frac'), e.get('verified'), e.get('error'), e.get('stats'))
PY
tail -3 "$OUT/bench_default.err"
