#!/bin/bash
# (the profiled part of scripts/gpu_final_r06.sh alone: kernel stats and PMC passes of the two roofline workloads, without the page-in pass in the profiled process)
# Round-6 measurement set (round 5's script with the round number changed, the SpGEMM PMC pass at scale 22 and the block / shard set-up lines added): rocprofv3 kernel stats of the default command, PMC traffic (separate passes) for the two roofline workloads,
# the bench lines of every BASELINE config (default + extras, scale 20, config 0, kron26 block, row blocks, BFS / SSSP loops, masked SpGEMM).
cd "$(dirname "$0")/.."
TAG=${1:-r06}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
pmc() {  # pmc <workload> : three PMC passes -> $OUT/pmc_traffic_<workload>.json
  wl=$1
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d="$OUT/pmc_${wl}_$(echo $c | tr ' ' '_')"
    timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python bench.py --workload $wl --steps 2 --warmup 3 --no-cpu-baseline --no-extra --no-warm-build > "$d.json" 2> "$d.err"; echo "pmc $wl [$c] rc=$?"
  done
  python - "$OUT" "$wl" <<'PY'
import csv, sys, glob, json, collections, os
out, wl = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, f"pmc_{wl}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if any(t in k for t in ("grb::k_mxv", "grb::k_x_image", "grb::k_long_init", "grb::k_long_compact", "grb::k_pack", "grb::k_rows_tail")):
            agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# the first two calls of a matrix run other kernels (plain path, layout build): per-launch averages of the steady-state kernels only
per_kernel = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if "k_mxv_pull" not in k and "k_mxv_seams" not in k}
for k, d in per_kernel.items():
    print("  ", k, {c: round(v, 1) for c, v in d.items()})
fetch = sum(d.get("FETCH_SIZE", 0.0) for d in per_kernel.values())
write = sum(d.get("WRITE_SIZE", 0.0) for d in per_kernel.values())
hit = sum(d.get("TCC_HIT_sum", 0.0) for d in per_kernel.values())
miss = sum(d.get("TCC_MISS_sum", 0.0) for d in per_kernel.values())
rec = {"workload": wl, "scale": 24, "round": 6, "kernels": sorted(per_kernel),
       "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "traffic_bytes_per_launch_uncorrected": (fetch + write) * 1024,
       "traffic_band_bytes": [(fetch + write) * 1024, (2 * fetch + write) * 1024],
       "tcc_hit_rate": hit / (hit + miss) if hit + miss else None, "tcc_miss_x_128_bytes": miss * 128, "per_kernel": per_kernel,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum in three separate passes with --kernel-trace "
                 "(scripts/gpu_final_r06.sh); per-launch averages summed over the steady-state kernels of one GrB_mxv call; KiB units; FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md section HBM (gfx950 reports half the bytes of 16-B/lane streaming reads); the doubling is "
                 "calibrated for the streamed arrays only: quote the BAND between the uncorrected and the corrected value (traffic_band_bytes), "
                 "not the corrected figure alone; tcc_miss_x_128_bytes = L2 misses x one 128-byte line, an independent estimate."}
json.dump(rec, open(os.path.join(out, f"pmc_traffic_{wl}.json"), "w"), indent=1)
PY
  find "$OUT" -name '*counter_collection.csv' -size +4M -delete; find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
}
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value','unit','ms_per_step','verified')}, (d.get('roofline') or {}).get('frac'))"; }
echo "== rocprofv3 stats of the default command (headline workload only)"; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$TAG" -o bench -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra --no-warm-build > $OLDPWD/$OUT/bench_default_under_rocprof.json 2> $OLDPWD/$OUT/prof.err ); cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv; grep -E "grb::k_mxv|grb::k_x_image|grb::k_long_init" "$OUT/bench_default_kernel_stats.csv" | cut -c1-150 | head -8
echo "== rocprofv3 stats of the BFS level step"; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_${TAG}_bfs" -o bench -- python $OLDPWD/bench.py --workload mxv_lor_land_masked --no-cpu-baseline --no-extra --no-warm-build > $OLDPWD/$OUT/bench_lor_land_under_rocprof.json 2>/dev/null ); cp $(find /tmp/prof_${TAG}_bfs -name '*kernel_stats.csv' | head -1) $OUT/bench_lor_land_kernel_stats.csv
echo "== PMC"; pmc mxv_min_plus_masked; pmc mxv_lor_land_masked
