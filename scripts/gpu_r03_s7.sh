#!/bin/bash
# round 3, GPU session 7: per-kernel times of the lean kernels, masked vs unmasked (does skipping the masked lanes' records pay?)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
kt() {  # tag env...
  tag=$1; shift
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python $OLDPWD/bench.py --steps 10 --no-extra --no-cpu-baseline ${BENCH_WL:-} > /dev/null 2>&1 )
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/kernel_stats_$tag.csv
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(t in r["Name"] for t in ("grb::k_mxv", "k_long_", "grb::k_x_image")) and int(r["Calls"]) > 2:
        print("   ", r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
{ echo "== lean, unmasked"; BENCH_WL="--workload mxv_min_plus" kt a GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5;
  echo "== lean, visited 0.9"; BENCH_WL="--visited 0.9" kt b GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5;
  echo "== lean, visited 0.1"; BENCH_WL="--visited 0.1" kt c GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5;
  echo "== default, visited 0.9"; BENCH_WL="--visited 0.9" kt d; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
