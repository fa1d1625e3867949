#!/bin/bash
# Round-2 probe: micro-benchmarks of the gather paths + the round-1 library's per-kernel times on the three mxv workloads.
TAG=${1:-r02a}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== ubench"; timeout 300 scripts/ubench/gather_rates > "$OUT/ubench.jsonl" 2> "$OUT/ubench.err"; echo "rc=$?"; cat "$OUT/ubench.jsonl"
echo "== bench --extra"; timeout 600 python bench.py --steps 20 --no-cpu-baseline --extra > "$OUT/bench_extra.json" 2> "$OUT/bench.err"; echo "rc=$?"; cut -c1-1500 "$OUT/bench_extra.json"
echo "== rocprofv3 stats (all three workloads)"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 20 --no-cpu-baseline --extra > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rc=$?"; grep -E "grb::" "$OUT/prof/bench_kernel_stats.csv" | cut -c1-200
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
