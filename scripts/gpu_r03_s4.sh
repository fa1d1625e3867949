#!/bin/bash
# round 3, GPU session 4: groups per wavefront of the tagged row-group kernel (build variants), per-kernel times
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
( timeout 900 python -m pytest tests/test_random_parity.py -x -q -m gpu -k "hot_cold or long_short_row or reductions_over" ) > $O/tests.log 2>&1
tail -2 $O/tests.log
GRB_SHORT_KERNEL=5 bash scripts/variants_ab.sh run > $O/tag_groups.txt 2>&1; cat $O/tag_groups.txt
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
{ echo "== short_kernel 5 (K=4)"; kt tg4 GRB_SHORT_KERNEL=5;
  echo "== K=8"; kt tg8 GRB_SHORT_KERNEL=5 GRB_MI355X_LIB=$PWD/build/variants/GRB_TAG_GROUPS=8/libgrb_mi355x.so;
  echo "== K=2"; kt tg2 GRB_SHORT_KERNEL=5 GRB_MI355X_LIB=$PWD/build/variants/GRB_TAG_GROUPS=2/libgrb_mi355x.so;
  echo "== bfs step, short_kernel 5"; BENCH_WL="--workload mxv_lor_land_masked" kt bfs GRB_SHORT_KERNEL=5;
  echo "== unmasked, short_kernel 5"; BENCH_WL="--workload mxv_min_plus" kt unm GRB_SHORT_KERNEL=5; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
