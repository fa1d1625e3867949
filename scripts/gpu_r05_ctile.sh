#!/bin/bash
# Round 5: k_mxv_ctile with UB units per thread and pass (1 = as in round 4, 2, 4 = shipped, 8)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_ctile; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4))"; }
for v in ct_ub1 ct_ub2 shipped ct_ub8; do
  lib=""; [ $v != shipped ] && lib="$PWD/build/variants/$v/libgrb_mi355x.so"
  GRB_MI355X_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra > "$OUT/head_$v.json" 2>/dev/null; line "$OUT/head_$v.json" "$v masked"
  GRB_MI355X_LIB=$lib timeout 300 python bench.py --workload mxv_min_plus --no-cpu-baseline --no-extra > "$OUT/unm_$v.json" 2>/dev/null; line "$OUT/unm_$v.json" "$v unmasked"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ct -o h -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra > /dev/null 2>&1 ); cp $(find /tmp/prof_ct -name '*kernel_stats.csv' | head -1) "$OUT/headline_kernel_stats.csv"; grep -E "grb::k_mxv|grb::k_long_init" "$OUT/headline_kernel_stats.csv" | cut -c1-130 | head -6
echo "== parity (split layouts, ordered)"; timeout 600 python -m pytest tests/test_vertex_order.py tests/test_random_parity.py -m gpu -x -q -k "not mxm" 2>&1 | tail -2
