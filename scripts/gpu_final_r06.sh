#!/bin/bash
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
echo "== bench default (headline + extras)"; ( time timeout 900 python bench.py ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; line $OUT/bench_default.json
echo "== unmasked min_plus scale 24"; timeout 600 python bench.py --workload mxv_min_plus --no-extra > "$OUT/bench_s24_unmasked.json" 2>/dev/null; line $OUT/bench_s24_unmasked.json
echo "== bench scale 20 (configs[1])"; timeout 600 python bench.py --scale 20 --workload mxv_min_plus > "$OUT/bench_s20_minplus.json" 2>/dev/null; line $OUT/bench_s20_minplus.json
echo "== config 0"; timeout 600 python bench.py --workload uniform_fp64 --steps 50 > "$OUT/bench_uniform_fp64.json" 2>/dev/null; line $OUT/bench_uniform_fp64.json
echo "== kron26 rank 0 of 8 (configs[4] shape)"; timeout 900 python bench.py --workload kron26 --block 0/8 > "$OUT/kron26_block0of8.json" 2>/dev/null; line $OUT/kron26_block0of8.json
echo "== row blocks of the scale-24 graph"; for w in 1 2 4 8; do timeout 600 python bench.py --block 0/$w --no-cpu-baseline --no-extra --no-warm-build 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'block': '0/$w', 'ms_per_step': d['ms_per_step'], 'verified': d['verified']}))"; done > "$OUT/block_times_s24.jsonl"; cat "$OUT/block_times_s24.jsonl"
echo "== the same blocks without the set-up step (natural-order layouts, rounds 1-5)"; for w in 2 4 8; do timeout 600 python bench.py --block 0/$w --no-shard-setup --no-cpu-baseline --no-extra --no-warm-build 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'block': '0/$w', 'shard_setup': False, 'ms_per_step': d['ms_per_step'], 'verified': d['verified'], 'ordered': d['stats'].get('ordered')}))"; done > "$OUT/block_times_s24_no_setup.jsonl"; cat "$OUT/block_times_s24_no_setup.jsonl"
timeout 900 python bench.py --workload kron26 --block 0/8 --no-shard-setup --no-cpu-baseline > "$OUT/kron26_block0of8_no_setup.json" 2>/dev/null; line $OUT/kron26_block0of8_no_setup.json
echo "== ranked labels (GrX_Matrix_hint_ranked): whole graph, blocks of a block-cyclic dealing, Kronecker-26 block"
for spec in "" "--block 0/2" "--block 0/4" "--block 0/8" "--block 7/8"; do
  timeout 600 python bench.py --ranked $spec --no-cpu-baseline --no-extra --no-warm-build 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'ranked': '$spec', 'ms_per_step': d['ms_per_step'], 'frac': d['roofline']['frac'], 'verified': d['verified'], 'ordered': d['stats'].get('ordered'), 'build_ms': d['layout_build_call_ms']}))"
done > "$OUT/ranked_block_times_s24.jsonl"; cat "$OUT/ranked_block_times_s24.jsonl"
timeout 900 python bench.py --workload kron26 --ranked --block 0/8 --no-cpu-baseline > "$OUT/kron26_ranked_block0of8.json" 2>/dev/null; line $OUT/kron26_ranked_block0of8.json
echo "== BFS / SSSP loops"; timeout 600 python bench.py --workload bfs --steps 5 > "$OUT/bfs_s24.json" 2>/dev/null; line $OUT/bfs_s24.json; timeout 600 python bench.py --workload sssp --steps 3 > "$OUT/sssp_s24.json" 2>/dev/null; line $OUT/sssp_s24.json
echo "== masked SpGEMM"; timeout 600 python bench.py --workload mxm_plus_times_masked --scale 20 --steps 5 --warmup 1 > "$OUT/mxm_masked_s20.json" 2>/dev/null; line $OUT/mxm_masked_s20.json; timeout 900 python bench.py --workload mxm_plus_times_masked --scale 22 --steps 3 --warmup 1 > "$OUT/mxm_masked_s22.json" 2>/dev/null; line $OUT/mxm_masked_s22.json
echo "== SpGEMM scale 18"; timeout 600 python bench.py --workload mxm_plus_times --scale 18 --steps 5 --warmup 1 > "$OUT/mxm_s18.json" 2>/dev/null; line $OUT/mxm_s18.json

echo "== PMC of the SpGEMM product (scale 20)"
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  d="$OUT/pmc_mxm_$(echo $c | tr ' ' '_')"
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python bench.py --workload mxm_plus_times --scale 20 --steps 1 --warmup 1 --no-cpu-baseline > "$d.json" 2> "$d.err"; echo "pmc mxm [$c] rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, json, os, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
calls = collections.defaultdict(set)
for f in glob.glob(os.path.join(out, "pmc_mxm_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "grb::" in k:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
# (2 products per run: warm-up + the timed one)
fetch, write = tot.get("FETCH_SIZE", 0.0) / 2, tot.get("WRITE_SIZE", 0.0) / 2
rec = {"workload": "mxm_plus_times", "scale": 20, "round": 6, "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "traffic_bytes_per_launch_uncorrected": (fetch + write) * 1024,
       "traffic_band_bytes": [(fetch + write) * 1024, (2 * fetch + write) * 1024],
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes with --kernel-trace; summed over every grb:: kernel of one product "
                 "(two products per run, halved); KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md: quote the band"}
json.dump(rec, open(os.path.join(out, "pmc_traffic_mxm.json"), "w"), indent=1)
print("mxm traffic band per product (GB):", [round(x / 1e9, 1) for x in rec["traffic_band_bytes"]])
PY
find "$OUT" -name '*counter_collection.csv' -size +4M -delete; find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
echo "== PMC of the SpGEMM product at scale 22 (streamed: the line the 1 -> 8 GPU target is quoted on)"
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  d="$OUT/pmc_mxm22_$(echo $c | tr ' ' '_')"
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- python bench.py --workload mxm_plus_times --scale 22 --steps 1 --warmup 1 --no-cpu-baseline > "$d.json" 2> "$d.err"; echo "pmc mxm22 [$c] rc=$?"
done
python - "$OUT" <<'PY'
import csv, sys, glob, json, os, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(os.path.join(out, "pmc_mxm22_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "grb::" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
fetch, write = tot.get("FETCH_SIZE", 0.0) / 2, tot.get("WRITE_SIZE", 0.0) / 2  # (2 products per run: warm-up + the timed one)
rec = {"workload": "mxm_plus_times_streamed", "scale": 22, "round": 6, "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "traffic_bytes_per_launch_uncorrected": (fetch + write) * 1024,
       "traffic_band_bytes": [(fetch + write) * 1024, (2 * fetch + write) * 1024],
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes with --kernel-trace over `bench.py --workload mxm_plus_times --scale 22` "
                 "(row batches, output streamed); summed over every grb:: kernel of one product (two products per run, halved); KiB units; FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md: quote the band"}
json.dump(rec, open(os.path.join(out, "pmc_traffic_mxm_s22.json"), "w"), indent=1)
print("mxm scale 22 traffic band per product (GB):", [round(x / 1e9, 1) for x in rec["traffic_band_bytes"]])
PY
find "$OUT" -name '*counter_collection.csv' -size +4M -delete; find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
echo "== the headline with round-5 / round-4 features off one at a time (same library, options)"
for kv in "GRB_ROWS_TILE=0" "GRB_ORDER_MODE=0" "GRB_VALUE_DICT=0" "GRB_HUB_MIN_LEN=0"; do
  env $kv timeout 600 python bench.py --no-cpu-baseline --no-extra --no-warm-build > "$OUT/bench_headline_$(echo $kv | tr ' =' '__').json" 2>/dev/null; echo -n "$kv: "; line "$OUT/bench_headline_$(echo $kv | tr ' =' '__').json"
done
echo "== SpGEMM kernel stats (scale 20 / 22, default options)"
for s in 20 22; do ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mxm$s -o mxm -- python $OLDPWD/bench.py --workload mxm_plus_times --scale $s --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 ); cp $(find /tmp/prof_mxm$s -name '*kernel_stats.csv' | head -1) "$OUT/mxm_s${s}_kernel_stats.csv"; head -7 "$OUT/mxm_s${s}_kernel_stats.csv" | cut -c1-150; done
echo "== RCCL one rank"; export HSA_ENABLE_IPC_MODE_LEGACY=0; timeout 600 python tests/nccl_one_rank.py > "$OUT/rccl_one_rank.json" 2> "$OUT/rccl_one_rank.err"; tail -c 400 "$OUT/rccl_one_rank.json"; timeout 600 python bench.py --force-dist --backend nccl --no-cpu-baseline > "$OUT/bench_force_dist_nccl_one_rank.json" 2>/dev/null; line "$OUT/bench_force_dist_nccl_one_rank.json"
echo "== layout build kernels"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_build -o b -- python $OLDPWD/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-extra --no-warm-build > /dev/null 2>&1 ); cp $(find /tmp/prof_build -name '*kernel_stats.csv' | head -1) "$OUT/layout_build_kernel_stats.csv"; head -25 "$OUT/layout_build_kernel_stats.csv" | cut -c1-120
echo "== pytest -m gpu (everything)"; ( time timeout 1200 python -m pytest tests -m gpu -q ) > "$OUT/tests_gpu_full.log" 2>&1; tail -4 "$OUT/tests_gpu_full.log"

