#!/bin/bash
# Round-2 measurement set: rocprofv3 kernel stats, PMC traffic (separate passes) for the two roofline workloads, bench lines
# (default + extras, scale 20, config 1, BFS / SSSP loops, SpGEMM plain / masked / streamed scale 22), first-call profile.
TAG=${1:-r02z}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
pmc() {  # pmc <workload> : three PMC passes -> $OUT/pmc_traffic_<workload>.json
  wl=$1
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d="$OUT/pmc_${wl}_$(echo $c | tr ' ' '_')"
    timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python bench.py --workload $wl --steps 2 --warmup 3 --no-cpu-baseline > "$d.json" 2> "$d.err"; echo "pmc $wl [$c] rc=$?"
  done
  python - "$OUT" "$wl" <<'PY'
import csv, sys, glob, json, collections, os
out, wl = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, f"pmc_{wl}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if any(t in k for t in ("grb::k_mxv", "grb::k_x_image", "grb::k_long_init", "grb::k_long_compact", "grb::k_pack")):
            agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# the first two calls of a matrix run other kernels (plain path, layout build): per-launch averages of the steady-state kernels only
per_kernel = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if "k_mxv_pull" not in k and "k_mxv_seams" not in k}
for k, d in per_kernel.items():
    print("  ", k, {c: round(v, 1) for c, v in d.items()})
fetch = sum(d.get("FETCH_SIZE", 0.0) for d in per_kernel.values())
write = sum(d.get("WRITE_SIZE", 0.0) for d in per_kernel.values())
hit = sum(d.get("TCC_HIT_sum", 0.0) for d in per_kernel.values())
miss = sum(d.get("TCC_MISS_sum", 0.0) for d in per_kernel.values())
rec = {"workload": wl, "scale": 24, "round": 2, "kernels": sorted(per_kernel),
       "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "traffic_bytes_per_launch_uncorrected": (fetch + write) * 1024,
       "tcc_hit_rate": hit / (hit + miss) if hit + miss else None, "per_kernel": per_kernel,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum in three separate passes with --kernel-trace "
                 "(scripts/gpu_final_r02.sh); per-launch averages summed over the steady-state kernels of one GrB_mxv call; KiB units; FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md section HBM (gfx950 reports half the bytes of 16-B/lane streaming reads); the doubling is "
                 "calibrated for the streamed arrays only: the true figure lies between the uncorrected and the corrected value."}
json.dump(rec, open(os.path.join(out, f"pmc_traffic_{wl}.json"), "w"), indent=1)
PY
  find "$OUT" -name '*counter_collection.csv' -size +4M -delete; find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
}
echo "== rocprofv3 stats (three mxv workloads)"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rc=$?"; grep -E "grb::" "$OUT/prof/bench_kernel_stats.csv" | cut -c1-170 | head -14
echo "== PMC"; pmc mxv_min_plus_masked; pmc mxv_lor_land_masked
mkdir -p profiles/r02; cp "$OUT/pmc_traffic_mxv_min_plus_masked.json" profiles/r02/pmc_traffic.json; cp "$OUT/pmc_traffic_mxv_lor_land_masked.json" profiles/r02/pmc_traffic_lor_land.json
echo "== bench default + extra"; timeout 900 python bench.py --steps 20 --warmup 5 --extra > "$OUT/bench_s24.json" 2> "$OUT/bench_s24.err"; echo "rc=$?"; cut -c1-500 "$OUT/bench_s24.json"
echo "== bench scale 20 (configs[1])"; timeout 600 python bench.py --scale 20 --steps 20 --warmup 5 --workload mxv_min_plus > "$OUT/bench_s20_minplus.json" 2> "$OUT/bench_s20.err"; echo "rc=$?"; cut -c1-300 "$OUT/bench_s20_minplus.json"
echo "== config 1"; timeout 600 python bench.py --workload uniform_fp64 --steps 50 > "$OUT/bench_uniform_fp64.json" 2> "$OUT/uni.err"; echo "rc=$?"; cut -c1-300 "$OUT/bench_uniform_fp64.json"
echo "== one rank's block of an 8-way run (compute only)"; for b in 0/1 0/2 0/4 0/8; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --block $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'block': '$b', 'ms_per_step': d['ms_per_step'], 'edges': d['config']['edges_counted_per_step'], 'verified': d['verified']}))"; done | tee "$OUT/block_times_s24.jsonl"
echo "== kron26 block 0/8"; timeout 900 python bench.py --workload kron26 --block 0/8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/kron26_block0of8.json" 2> "$OUT/kron26.err"; echo "rc=$?"; cut -c1-400 "$OUT/kron26_block0of8.json"
echo "== bfs"; for s in 20 24; do timeout 600 python bench.py --workload bfs --scale $s --steps 5 --warmup 2 > "$OUT/bfs_s$s.json" 2> "$OUT/bfs_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/bfs_s$s.json"; done
echo "== sssp"; for s in 20 24; do timeout 600 python bench.py --workload sssp --scale $s --steps 3 --warmup 1 > "$OUT/sssp_s$s.json" 2> "$OUT/sssp_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/sssp_s$s.json"; done
echo "== mxm masked"; for s in 20 22; do timeout 900 python bench.py --workload mxm_plus_times_masked --scale $s --steps 2 --warmup 1 > "$OUT/mxm_masked_s$s.json" 2> "$OUT/mxm_masked_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/mxm_masked_s$s.json"; done
echo "== mxm"; for s in 18 20; do timeout 900 python bench.py --workload mxm_plus_times --scale $s --steps 2 --warmup 1 > "$OUT/mxm_s$s.json" 2> "$OUT/mxm_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/mxm_s$s.json"; done
echo "== mxm scale 22 streamed"; timeout 900 python bench.py --workload mxm_plus_times --scale 22 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/mxm_s22_streamed.json" 2> "$OUT/mxm_s22.err"; echo "rc=$?"; cut -c1-500 "$OUT/mxm_s22_streamed.json"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_mxm" -o mxm -- python bench.py --workload mxm_plus_times --scale 20 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/prof_mxm.json" 2> "$OUT/prof_mxm.err"; grep -E "grb::" "$OUT/prof_mxm/mxm_kernel_stats.csv" | cut -c1-160 | head -8
echo "== first call"; bash scripts/gpu_first_call.sh $TAG/first | head -4
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete; find "$OUT" -name '*counter_collection.csv' -size +4M -delete
