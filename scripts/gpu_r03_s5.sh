#!/bin/bash
# round 3, GPU session 5: lean hot / cold strip kernels (k_mxv_hstrip / k_mxv_cstrip, long_kernel = 5) + tagged row groups
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
( timeout 900 python -m pytest tests/test_random_parity.py -x -q -m gpu -k "hot_cold or long_short_row or many_chunks" ) > $O/tests.log 2>&1
tail -2 $O/tests.log
one() {  # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'launches', d['stats']['kernel_launches'])"
}
{
one default
one lean_hc16 GRB_LONG_KERNEL=5
one lean_hc16_tag GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5
one lean_hc32_tag GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_LONG_CLASSES=32
one lean_hc16_tag_len32 GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_SPLIT_MIN_LEN=32
one lean_hc16_tag_len128 GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_SPLIT_MIN_LEN=128
one lean_hc32_tag_len32 GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_LONG_CLASSES=32 GRB_SPLIT_MIN_LEN=32
BENCH_WL="--workload mxv_min_plus" one unmasked_default
BENCH_WL="--workload mxv_min_plus" one unmasked_lean GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5
} > $O/ab.txt 2>&1
cat $O/ab.txt
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
{ echo "== long 5 short 5, 16 classes"; kt a GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5;
  echo "== long 5 short 5, 32 classes"; kt b GRB_LONG_KERNEL=5 GRB_SHORT_KERNEL=5 GRB_LONG_CLASSES=32; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
