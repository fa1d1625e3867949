#!/bin/bash
# round 3, GPU session 6: deeper record prefetch in k_mxv_hstrip; hot tables beyond the LDS-resident codes (a warm, L2-resident tier)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
one() {  # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'hot_k', d['stats']['hot_k'])"
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
one lean GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5
one lean_hot1M GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_HOT_K=1048576
one lean_hot2M GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_HOT_K=2097152
one lean_hot4M GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_HOT_K=4194304
} > $O/ab.txt 2>&1
cat $O/ab.txt
{ echo "== lean"; kt a GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5;
  echo "== lean hot_k 2M"; kt b GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_HOT_K=2097152; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
