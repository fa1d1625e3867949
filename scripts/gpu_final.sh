#!/bin/bash
# Round-end measurement set: bench lines, rocprofv3 kernel stats, PMC counters (separate passes), mxm lines.
TAG=${1:-r01z}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== bench default + extra"; timeout 900 python bench.py --steps 20 --extra > "$OUT/bench_s24.json" 2> "$OUT/bench_s24.err"; echo "rc=$?"; cut -c1-600 "$OUT/bench_s24.json"
echo "== bench scale 20 (configs[1])"; timeout 600 python bench.py --scale 20 --steps 20 --workload mxv_min_plus > "$OUT/bench_s20_minplus.json" 2> "$OUT/bench_s20.err"; echo "rc=$?"; cut -c1-400 "$OUT/bench_s20_minplus.json"
echo "== rocprofv3 stats"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 20 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rc=$?"; grep -E "grb::" "$OUT/prof/bench_kernel_stats.csv" | cut -c1-160
echo "== pmc FETCH_SIZE"; timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"; echo "rc=$?"
echo "== pmc WRITE_SIZE"; timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"; echo "rc=$?"
echo "== pmc TCC hit/miss"; timeout 420 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_tcc.json" 2> "$OUT/pmc_tcc.err"; echo "rc=$?"
for d in pmc_fetch pmc_write pmc_tcc; do f=$(find "$OUT/$d" -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "grb::" in k:
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
PY
done
echo "== mxm"; for s in 18 20; do timeout 900 python bench.py --workload mxm_plus_times --scale $s --steps 2 --warmup 1 > "$OUT/mxm_s$s.json" 2> "$OUT/mxm_s$s.err"; echo "rc=$?"; cut -c1-500 "$OUT/mxm_s$s.json"; done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_mxm" -o mxm -- python bench.py --workload mxm_plus_times --scale 18 --steps 2 --warmup 1 > "$OUT/prof_mxm.json" 2> "$OUT/prof_mxm.err"; grep -E "grb::" "$OUT/prof_mxm/mxm_kernel_stats.csv" | cut -c1-160
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete; find "$OUT" -name '*counter_collection.csv' -size +8M -delete
