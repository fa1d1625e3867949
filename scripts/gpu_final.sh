#!/bin/bash
# Round-end measurement set: bench lines, rocprofv3 kernel stats, PMC counters (separate passes), mxm lines.
TAG=${1:-r01z}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== rocprofv3 stats"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 20 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rc=$?"; grep -E "grb::" "$OUT/prof/bench_kernel_stats.csv" | cut -c1-160
echo "== pmc FETCH_SIZE"; timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"; echo "rc=$?"
echo "== pmc WRITE_SIZE"; timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"; echo "rc=$?"
echo "== pmc TCC hit/miss"; timeout 420 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_tcc.json" 2> "$OUT/pmc_tcc.err"; echo "rc=$?"
python - "$OUT" <<'PY'
# per-launch averages of the PMC counters for the kernels of one GrB_mxv call -> pmc_traffic.json
import csv, sys, glob, json, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_tcc"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if any(t in k for t in ("grb::k_mxv", "grb::k_x_image", "grb::k_long_init", "grb::k_long_compact")):  # the per-call kernels
                agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
per_kernel = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in per_kernel.items():
    print(k, {c: round(v, 1) for c, v in d.items()})
fetch = sum(d.get("FETCH_SIZE", 0.0) for d in per_kernel.values())
write = sum(d.get("WRITE_SIZE", 0.0) for d in per_kernel.values())
hit = sum(d.get("TCC_HIT_sum", 0.0) for d in per_kernel.values())
miss = sum(d.get("TCC_MISS_sum", 0.0) for d in per_kernel.values())
rec = {"workload": "mxv_min_plus_masked", "scale": 24, "kernels": sorted(per_kernel),
       "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "traffic_bytes_per_launch_uncorrected": (fetch + write) * 1024,
       "tcc_hit_rate": hit / (hit + miss) if hit + miss else None, "per_kernel": per_kernel,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum in three separate passes with "
                 "--kernel-trace (scripts/gpu_final.sh); per-launch averages summed over the kernels of one GrB_mxv call; KiB units; "
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM (gfx950 reports half the bytes of 16-B/lane streaming reads). "
                 "The doubling is exact for the streamed arrays only; the 4-byte x gathers are uncalibrated, so the true figure lies "
                 "between the uncorrected and the corrected value."}
json.dump(rec, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
PY
# the bench line reports roofline.traffic from profiles/r01/pmc_traffic.json: refresh it first (same kernels, this box)
cp "$OUT/pmc_traffic.json" profiles/r01/pmc_traffic.json
echo "== bench default + extra"; timeout 900 python bench.py --steps 20 --extra > "$OUT/bench_s24.json" 2> "$OUT/bench_s24.err"; echo "rc=$?"; cut -c1-600 "$OUT/bench_s24.json"
echo "== bench scale 20 (configs[1])"; timeout 600 python bench.py --scale 20 --steps 20 --workload mxv_min_plus > "$OUT/bench_s20_minplus.json" 2> "$OUT/bench_s20.err"; echo "rc=$?"; cut -c1-400 "$OUT/bench_s20_minplus.json"
echo "== bfs"; for s in 20 24; do timeout 600 python bench.py --workload bfs --scale $s --steps 5 --warmup 1 > "$OUT/bfs_s$s.json" 2> "$OUT/bfs_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/bfs_s$s.json"; done
echo "== sssp"; for s in 20 24; do timeout 600 python bench.py --workload sssp --scale $s --steps 3 --warmup 1 > "$OUT/sssp_s$s.json" 2> "$OUT/sssp_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/sssp_s$s.json"; done
echo "== mxm masked (C<A.S> = A A)"; for s in 20 22; do timeout 900 python bench.py --workload mxm_plus_times_masked --scale $s --steps 2 --warmup 1 > "$OUT/mxm_masked_s$s.json" 2> "$OUT/mxm_masked_s$s.err"; echo "rc=$?"; cut -c1-500 "$OUT/mxm_masked_s$s.json"; done
echo "== mxm"; for s in 18 20; do timeout 900 python bench.py --workload mxm_plus_times --scale $s --steps 2 --warmup 1 > "$OUT/mxm_s$s.json" 2> "$OUT/mxm_s$s.err"; echo "rc=$?"; cut -c1-500 "$OUT/mxm_s$s.json"; done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_mxm" -o mxm -- python bench.py --workload mxm_plus_times --scale 18 --steps 2 --warmup 1 > "$OUT/prof_mxm.json" 2> "$OUT/prof_mxm.err"; grep -E "grb::" "$OUT/prof_mxm/mxm_kernel_stats.csv" | cut -c1-160
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete; find "$OUT" -name '*counter_collection.csv' -size +8M -delete
