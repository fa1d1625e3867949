#!/bin/bash
# Round 4, GPU call 3: the ordered layouts on the GPU -- parity tests of the new path, then the headline with / without the order,
# kernel stats and L2 counters of the ordered run.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_s3; mkdir -p "$OUT"; export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python -m pytest tests/test_vertex_order.py -x -q -m gpu 2>&1 | tail -5
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified','layout_build_call_ms','first_call_ms','preprocess_bytes')}, (d.get('roofline') or {}).get('frac'))"; }
for wl in mxv_min_plus_masked mxv_min_plus; do
  for om in 1 0; do
    GRB_ORDER_MODE=$om GRB_PRINT_STRIPS=1 timeout 600 python bench.py --workload $wl --no-extra --no-cpu-baseline > "$OUT/bench_${wl}_order$om.json" 2> "$OUT/bench_${wl}_order$om.err"
    line "$OUT/bench_${wl}_order$om.json" "$wl order_mode=$om"
  done
done
grep -h "strips kind\|cold tiles" $OUT/bench_mxv_min_plus_masked_order1.err | cut -c1-900
run() {  # run <tag> <bench args>
  tag=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$tag" -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-extra "$@" > $ROOT/$OUT/prof_bench_$tag.json 2> $ROOT/$OUT/prof_$tag.err )
  cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$tag.csv
  echo "== $tag"
  grep -E "grb::" "$OUT/kernel_stats_$tag.csv" | grep -v "pull\|seams" | head -30 | awk -F'",' '{split($2,a,","); printf "   %-60s calls %s avg %s us\n", substr($1,2,60), a[1], a[3]/1000}'
}
run masked
run unmasked --workload mxv_min_plus
d="$OUT/pmc_masked"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$d" -o p -- python bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2> "$d.err"
python - "$d" <<'PY'
import csv, sys, glob, os, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "grb::k_mxv" in k and "pull" not in k and "seams" not in k:
            agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    h = sum(d["TCC_HIT_sum"]) / len(d["TCC_HIT_sum"]); m = sum(d["TCC_MISS_sum"]) / len(d["TCC_MISS_sum"])
    print("  ", k, "hit rate %.3f" % (h / (h + m)), "miss bytes %.3f GB" % (m * 128 / 1e9))
PY
find "$OUT" -name '*counter_collection.csv' -size +4M -delete; find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
