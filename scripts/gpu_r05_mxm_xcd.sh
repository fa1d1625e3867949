#!/bin/bash
# Round 5: SpGEMM unit order per XCD (mxm_xcd_map) x window groups; rtile defaults (1024 threads, 6 waves) sanity
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_mxm_xcd; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4))"; }
timeout 300 python bench.py --no-cpu-baseline --no-extra > "$OUT/head.json" 2>/dev/null; line "$OUT/head.json" "headline (rtile 1024 threads)"
for s in 22 20; do for x in 0 1; do for f in 1 2 4; do
  GRB_MXM_XCD_MAP=$x GRB_MXM_WINDOW_GROUPS=$f timeout 600 python bench.py --workload mxm_plus_times --scale $s --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/mxm_s${s}_x${x}_g$f.json" 2> "$OUT/mxm_s${s}_x${x}_g$f.err"; line "$OUT/mxm_s${s}_x${x}_g$f.json" "s$s xcd_map=$x groups=$f"
done; done; done
echo "== mxm parity on the GPU with the map on"; timeout 900 python -m pytest tests -m gpu -x -q -k "mxm" 2>&1 | tail -2
