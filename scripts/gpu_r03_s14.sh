#!/bin/bash
# round 3, GPU session 14: the cold entries of the long rows as tagged tiles (k_mxv_ctile)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s14; mkdir -p $O
( timeout 900 python -m pytest tests/test_random_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "hot_cold or cold_tiles or long_short_row or many_chunks or scale24_headline or rmat_vs_oracle or row_block" ) > $O/tests.log 2>&1
tail -3 $O/tests.log
one() {  # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'build_ms', round(d['layout_build_call_ms'],1))"
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
one sub1 GRB_LONG_SUB=1
one sub2 GRB_LONG_SUB=2
one sub8 GRB_LONG_SUB=8
one c32 GRB_LONG_CLASSES=32
BENCH_WL="--workload mxv_min_plus" one unmasked
} > $O/ab.txt 2>&1
cat $O/ab.txt
{ echo "== default"; kt a; echo "== unmasked"; BENCH_WL="--workload mxv_min_plus" kt b; } > $O/kernel_times.txt 2>&1
cat $O/kernel_times.txt
