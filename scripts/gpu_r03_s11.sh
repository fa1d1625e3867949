#!/bin/bash
# round 3, GPU session 11: a warm tier -- hot tables larger than the LDS-resident codes, the extra codes dealt to the cold classes by line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
one() {  # tag env...
  tag=$1; shift
  env GRB_PRINT_STRIPS=1 "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>$O/err_$tag.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'hot_k', d['stats']['hot_k'])"
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
one hot1M GRB_HOT_K=1048576
one hot1.5M GRB_HOT_K=1572864
one hot2M GRB_HOT_K=2097152
one hot3M GRB_HOT_K=3145728
BENCH_WL="--workload mxv_min_plus" one unmasked_default
BENCH_WL="--workload mxv_min_plus" one unmasked_hot1.5M GRB_HOT_K=1572864
} > $O/ab.txt 2>&1
cat $O/ab.txt
{ echo "== hot_k 1.5M"; kt a GRB_HOT_K=1572864; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
