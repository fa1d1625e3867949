#!/bin/bash
# Round 2, SpGEMM measurement set after the (row, window) unit kernels: rocprofv3 kernel stats and PMC traffic (separate
# passes) of the scale-20 product, bench lines (scale 18 / 20 with the CPU baseline, scale 22 streamed, masked 20 / 22).
TAG=${1:-r02mxm}; OUT=gpurun_out/$TAG; mkdir -p "$OUT" profiles/r02; export TMPDIR=/tmp
STEPS=2; WARM=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_mxm" -o mxm -- python bench.py --workload mxm_plus_times --scale 20 --steps $STEPS --warmup $WARM --no-cpu-baseline > "$OUT/prof_mxm.json" 2> "$OUT/prof_mxm.err"
echo "stats rc=$?"; grep -E "grb::" "$OUT/prof_mxm/mxm_kernel_stats.csv" | cut -c1-160 | head -12
cp "$OUT/prof_mxm/mxm_kernel_stats.csv" profiles/r02/mxm_s20_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  d="$OUT/pmc_mxm_$c"
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python bench.py --workload mxm_plus_times --scale 20 --steps $STEPS --warmup $WARM --no-cpu-baseline > "$d.json" 2> "$d.err"; echo "pmc mxm [$c] rc=$?"
done
python - "$OUT" $((STEPS + WARM)) <<'PY'
import csv, sys, glob, json, collections, os
out, calls = sys.argv[1], int(sys.argv[2])
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "pmc_mxm_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "grb::" in k or "rocprim" in k:
            tot[k.split("(")[0][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
per_kernel = {k: {c: v / calls for c, v in d.items()} for k, d in tot.items()}
fetch = sum(d.get("FETCH_SIZE", 0.0) for d in per_kernel.values())
write = sum(d.get("WRITE_SIZE", 0.0) for d in per_kernel.values())
for k, d in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))[:10]:
    print("  ", k, {c: round(v / 1e6, 2) for c, v in d.items()}, "(GiB)")
rec = {"workload": "mxm_plus_times", "scale": 20, "round": 2, "kernels": sorted(per_kernel),
       "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "traffic_bytes_per_launch_uncorrected": (fetch + write) * 1024,
       "per_kernel": per_kernel,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes with --kernel-trace (scripts/gpu_r02_mxm_final.sh); "
                 "summed over every kernel of one GrB_mxm call (all launches of the run / the number of calls); KiB units; FETCH_SIZE doubled "
                 "per MI355X_MICROARCH.md section HBM (calibrated for streaming reads: the true figure lies between the uncorrected and "
                 "the corrected value)."}
json.dump(rec, open(os.path.join(out, "pmc_traffic_mxm.json"), "w"), indent=1)
print("traffic GB per call (corrected / uncorrected):", rec["traffic_bytes_per_launch"] / 1e9, rec["traffic_bytes_per_launch_uncorrected"] / 1e9)
PY
cp "$OUT/pmc_traffic_mxm.json" profiles/r02/pmc_traffic_mxm.json
find "$OUT" -name '*counter_collection.csv' -size +4M -delete; find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
echo "== mxm"; for s in 18 20; do timeout 900 python bench.py --workload mxm_plus_times --scale $s --steps 3 --warmup 1 > "$OUT/mxm_s$s.json" 2> "$OUT/mxm_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/mxm_s$s.json"; done
echo "== mxm masked"; for s in 20 22; do timeout 900 python bench.py --workload mxm_plus_times_masked --scale $s --steps 2 --warmup 1 > "$OUT/mxm_masked_s$s.json" 2> "$OUT/mxm_masked_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/mxm_masked_s$s.json"; done
echo "== mxm scale 22 streamed"; timeout 1200 python bench.py --workload mxm_plus_times --scale 22 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/mxm_s22_streamed.json" 2> "$OUT/mxm_s22.err"; echo "rc=$?"; cut -c1-500 "$OUT/mxm_s22_streamed.json"
for f in mxm_s18 mxm_s20 mxm_masked_s20 mxm_masked_s22 mxm_s22_streamed; do [ -s "$OUT/$f.json" ] && cp "$OUT/$f.json" profiles/r02/$f.json; done
cp -r profiles/r02 "$OUT/profiles_r02"
