#!/bin/bash
# round 3, GPU session 12: the BFS level step (configs[2]): predicated LDS atomics in the tagged row groups; long-row layouts for BOOL
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
one() {  # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'long_kernel', d['stats']['long_kernel'])"
}
kt() {  # tag env...
  tag=$1; shift
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python $OLDPWD/bench.py --steps 10 --no-extra --no-cpu-baseline ${BENCH_WL:-} > /dev/null 2>&1 )
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/kernel_stats_$tag.csv
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(t in r["Name"] for t in ("grb::k_mxv", "k_long_", "grb::k_x_image", "k_pack")) and int(r["Calls"]) > 2:
        print("   ", r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
export BENCH_WL="--workload mxv_lor_land_masked"
{
one bfs_default
one bfs_hotcold GRB_LONG_KERNEL=4
one bfs_strips GRB_LONG_KERNEL=2
one bfs_rows1 GRB_SHORT_KERNEL=1
} > $O/ab.txt 2>&1
cat $O/ab.txt
{ echo "== bfs default"; kt a; echo "== bfs hot/cold strips"; kt b GRB_LONG_KERNEL=4; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
