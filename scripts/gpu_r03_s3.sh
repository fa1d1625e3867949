#!/bin/bash
# round 3, GPU session 3: tagged row groups (short_kernel = 5) against the row-group kernel: parity, A/B, per-kernel times
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
( timeout 900 python -m pytest tests/test_random_parity.py -x -q -m gpu -k "hot_cold or long_short_row or reductions_over" ) > $O/tests.log 2>&1
tail -3 $O/tests.log
one() {  # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'build_ms', round(d['layout_build_call_ms'],1), 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'launches', d['stats']['kernel_launches'])"
}
{
one default
one tagged GRB_SHORT_KERNEL=5
BENCH_WL="--workload mxv_lor_land_masked" one bfs_default
BENCH_WL="--workload mxv_lor_land_masked" one bfs_tagged GRB_SHORT_KERNEL=5
BENCH_WL="--workload mxv_min_plus" one unmasked_default
BENCH_WL="--workload mxv_min_plus" one unmasked_tagged GRB_SHORT_KERNEL=5
one default
} > $O/ab.txt 2>&1
cat $O/ab.txt
kt() {  # tag env...
  tag=$1; shift
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python $OLDPWD/bench.py --steps 10 --no-extra --no-cpu-baseline > /dev/null 2>&1 )
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/kernel_stats_$tag.csv
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(t in r["Name"] for t in ("grb::k_mxv", "k_long_", "grb::k_x_image", "k_tag")):
        print("   ", r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
{ echo "== short_kernel 5"; kt tg GRB_SHORT_KERNEL=5; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
