#!/bin/bash
# Round 4: rocprofv3 kernel stats of the headline (masked), the unmasked relaxation and the BFS level step; prints the steady-state kernels.
#   bash scripts/gpu_r04_kstats.sh <tag> [workloads...]      (GRB_* environment knobs pass through)
cd "$(dirname "$0")/.."
TAG=${1:-kstats}; shift
WLS=${@:-mxv_min_plus_masked mxv_min_plus mxv_lor_land_masked}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
ROOT=$PWD
for wl in $WLS; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_${TAG}_$wl" -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-extra --workload $wl > $ROOT/$OUT/bench_$wl.json 2> $ROOT/$OUT/prof_$wl.err )
  cp $(find /tmp/prof_${TAG}_$wl -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$wl.csv
  python - "$OUT/bench_$wl.json" "$wl" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("==", sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "verified", "layout_build_call_ms", "preprocess_bytes")}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("==", sys.argv[2], "FAILED", e)
PY
  grep -E "grb::" "$OUT/kernel_stats_$wl.csv" | awk -F'",' '{split($2,a,","); if (a[1] >= 20) printf "   %-64s calls %s avg %.1f us\n", substr($1,2,64), a[1], a[3]/1000}'
done
