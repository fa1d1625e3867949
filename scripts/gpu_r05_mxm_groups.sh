#!/bin/bash
# Round 5, first GPU call: the SpGEMM units over GROUPS of column windows (option mxm_window_groups) and the checksum folded into the
# numeric stores -- parity tests of the product on the GPU, then the sweep at scale 20 / 22 (and 18 / 21 for the default policy).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_mxm; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4))"; }
echo "== pytest -m gpu -k mxm"; ( time timeout 900 python -m pytest tests -m gpu -x -q -k "mxm or gxb or view_held" ) > "$OUT/tests_mxm.log" 2>&1; tail -3 "$OUT/tests_mxm.log"
for s in 20 22; do
  for f in 1 2 4 8 0; do
    GRB_MXM_WINDOW_GROUPS=$f timeout 600 python bench.py --workload mxm_plus_times --scale $s --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/mxm_s${s}_groups$f.json" 2> "$OUT/mxm_s${s}_groups$f.err"; line "$OUT/mxm_s${s}_groups$f.json" "s$s groups=$f"
  done
done
GRB_MXM_CHECKSUM_PASS=1 timeout 600 python bench.py --workload mxm_plus_times --scale 22 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/mxm_s22_checksum_pass.json" 2>/dev/null; line "$OUT/mxm_s22_checksum_pass.json" "s22 default groups, checksum by its own pass"
for s in 18 21; do timeout 600 python bench.py --workload mxm_plus_times --scale $s --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/mxm_s$s.json" 2>/dev/null; line "$OUT/mxm_s$s.json" "s$s default"; done
GRB_MXM_WINDOW_GROUPS=1 timeout 600 python bench.py --workload mxm_plus_times --scale 21 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/mxm_s21_groups1.json" 2>/dev/null; line "$OUT/mxm_s21_groups1.json" "s21 groups=1"
echo "== rocprofv3 kernel stats, scale 22, default groups"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mxm22 -o mxm -- python $OLDPWD/bench.py --workload mxm_plus_times --scale 22 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 ); cp $(find /tmp/prof_mxm22 -name '*kernel_stats.csv' | head -1) "$OUT/mxm_s22_kernel_stats.csv"; head -12 "$OUT/mxm_s22_kernel_stats.csv" | cut -c1-160
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mxm20 -o mxm -- python $OLDPWD/bench.py --workload mxm_plus_times --scale 20 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 ); cp $(find /tmp/prof_mxm20 -name '*kernel_stats.csv' | head -1) "$OUT/mxm_s20_kernel_stats.csv"; head -12 "$OUT/mxm_s20_kernel_stats.csv" | cut -c1-160
