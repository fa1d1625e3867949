#!/bin/bash
# round 3, GPU session 2: hot / cold strips (long_kernel = 5) against the class strips -- parity on the GPU, A/B of the headline
# call over class counts and long-row thresholds, per-kernel times
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
( timeout 900 python -m pytest tests/test_random_parity.py -x -q -m gpu -k "hot_cold or long_short_row or many_chunks or terminal or split_survives" ) > $O/tests.log 2>&1
tail -3 $O/tests.log
one() {  # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'build_ms', round(d['layout_build_call_ms'],1), 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'launches', d['stats']['kernel_launches'], 'segs', d['stats']['long_segments'])"
}
{
one default GRB_LONG_KERNEL=3
one hotcold_c16 GRB_LONG_KERNEL=5
one hotcold_c32 GRB_LONG_KERNEL=5 GRB_LONG_CLASSES=32
one hotcold_c16_len32 GRB_LONG_KERNEL=5 GRB_SPLIT_MIN_LEN=32
one hotcold_c16_len128 GRB_LONG_KERNEL=5 GRB_SPLIT_MIN_LEN=128
one hotcold_c32_len128 GRB_LONG_KERNEL=5 GRB_LONG_CLASSES=32 GRB_SPLIT_MIN_LEN=128
one hotcold_c8 GRB_LONG_KERNEL=5 GRB_LONG_CLASSES=8
one default GRB_LONG_KERNEL=3
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
    if any(t in r["Name"] for t in ("grb::k_mxv", "k_long_", "grb::k_x_image")):
        print("   ", r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
{ echo "== long_kernel 5"; kt hc GRB_LONG_KERNEL=5; echo "== long_kernel 5, 32 classes"; kt hc32 GRB_LONG_KERNEL=5 GRB_LONG_CLASSES=32; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
