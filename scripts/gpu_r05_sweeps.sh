#!/bin/bash
# Round 5, late: small sweeps -- k_mxv_rtile wavefronts per SIMD at 1024 threads (build variants), BOOL tile size, SpGEMM options at scale 22
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_sweeps; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4))"; }
for v in shipped rt_b1024_w8 rt_b1024_w4; do
  lib=""; [ $v != shipped ] && lib="$PWD/build/variants/$v/libgrb_mi355x.so"
  GRB_MI355X_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extra > "$OUT/head_$v.json" 2>/dev/null; line "$OUT/head_$v.json" "$v masked"
done
timeout 300 python bench.py --workload mxv_lor_land_masked --no-cpu-baseline --no-extra > "$OUT/bfs_step.json" 2>/dev/null; line "$OUT/bfs_step.json" "BFS step (BOOL tiles of 64 Ki entries)"
M="python bench.py --workload mxm_plus_times --scale 22 --steps 1 --warmup 1 --no-cpu-baseline"
for kv in "GRB_MXM_SYM_WINDOWS=4" "GRB_MXM_SYM_WINDOWS=16" "GRB_MXM_UNIT_SMALL=256" "GRB_MXM_BITMAP_MIN_CNT=256" "GRB_MXM_BITMAP_MIN_CNT=1024" "GRB_MXM_UNIT_MIN_PER_WINDOW=8" "GRB_MXM_UNIT_MIN_PER_WINDOW=32" "GRB_MXM_UNIT_MID=2048" "GRB_MXM_UNIT_DENSE=8192"; do
  env $kv timeout 300 $M > "$OUT/mxm_s22_$(echo $kv | tr ' =' '__').json" 2>/dev/null; line "$OUT/mxm_s22_$(echo $kv | tr ' =' '__').json" "s22 $kv"
done
timeout 300 $M > "$OUT/mxm_s22_default.json" 2>/dev/null; line "$OUT/mxm_s22_default.json" "s22 default"
