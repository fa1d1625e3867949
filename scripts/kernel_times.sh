#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) of the mxv workloads, one process each.
TAG=${1:-kt}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for wl in ${WLS:-mxv_min_plus_masked mxv_min_plus mxv_lor_land_masked}; do
  for fl in ${FLAGS:-0}; do
    GRB_DEBUG_FLAGS=$fl timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$wl.$fl" -o b -- python bench.py --workload $wl --steps 10 --no-cpu-baseline > "$OUT/$wl.$fl.json" 2> "$OUT/$wl.$fl.err"
    echo "== $wl flags=$fl rc=$? $(python -c "import json,sys; d=json.loads(open('$OUT/$wl.$fl.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
    python - "$OUT/$wl.$fl/b_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "grb::k_mxv" in r["Name"] or "k_long_" in r["Name"] or "grb::k_x_image" in r["Name"] or "rocclr" in r["Name"] or "k_fill_w" in r["Name"] or "k_vec_write" in r["Name"]:
        print("   ", r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
  done
done
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
