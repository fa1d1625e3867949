#!/bin/bash
# Round 4, GPU call 1: (a) does label locality pay with the round-3 kernels?  the headline on the graph as it is and relabeled three ways;
# (b) rocprofv3 kernel stats + L2 counters for "none" and "degree".
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_s1; mkdir -p "$OUT"; export TMPDIR=/tmp
for m in none degree colcount longfirst; do
  for wl in mxv_min_plus_masked mxv_min_plus mxv_lor_land_masked; do
    GRB_RELABEL=$m timeout 600 python scripts/relabel_probe.py --workload $wl --no-extra --no-cpu-baseline > "$OUT/bench_${m}_${wl}.json" 2> "$OUT/bench_${m}_${wl}.err"
    python - "$OUT/bench_${m}_${wl}.json" $m $wl <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], {k: d.get(k) for k in ("ms_per_step", "verified", "layout_build_call_ms")}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
for m in none degree; do
  ( cd /tmp && GRB_RELABEL=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$m" -o bench -- python $OLDPWD/scripts/relabel_probe.py --no-cpu-baseline --no-extra > /dev/null 2> $OLDPWD/$OUT/prof_$m.err )
  cp $(find /tmp/prof_$m -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$m.csv
  echo "== kernel stats $m"; grep -E "grb::k_mxv|grb::k_x_image|grb::k_long_init" "$OUT/kernel_stats_$m.csv" | cut -c1-200 | head -8
  d="$OUT/pmc_${m}"
  GRB_RELABEL=$m timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$d" -o p -- python scripts/relabel_probe.py --steps 2 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2> "$d.err"
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
done
