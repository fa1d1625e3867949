#!/bin/bash
# round 3, GPU session 10: hot lines dealt boustrophedon to the classes (class balance), both layouts
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
( timeout 900 python -m pytest tests/test_random_parity.py -x -q -m gpu -k "hot_cold or long_short_row or many_chunks or terminal" ) > $O/tests.log 2>&1
tail -2 $O/tests.log
one() {  # tag env...
  tag=$1; shift
  env GRB_PRINT_STRIPS=1 "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>$O/err_$tag.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'segs', d['stats']['long_segments'])"
  grep strips $O/err_$tag.txt | head -1
}
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
{
one default
one default_tagrows GRB_SHORT_KERNEL=5
one lean GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5
one lean_c32 GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_LONG_CLASSES=32
one lean_len128 GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_SPLIT_MIN_LEN=128
one lean_len256 GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_SPLIT_MIN_LEN=256
BENCH_WL="--workload mxv_min_plus" one unmasked_default
BENCH_WL="--workload mxv_min_plus" one unmasked_lean GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5
BENCH_WL="--workload mxv_lor_land_masked" one bfs_default
BENCH_WL="--workload mxv_lor_land_masked" one bfs_tagrows GRB_SHORT_KERNEL=5
} > $O/ab.txt 2>&1
cat $O/ab.txt
{ echo "== default"; kt a;
  echo "== lean"; kt b GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
