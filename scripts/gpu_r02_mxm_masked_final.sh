#!/bin/bash
# Round 2: GPU test tier + the mask-driven SpGEMM bench lines after the masked unit kernels
OUT=gpurun_out/r02mm; mkdir -p "$OUT" profiles/r02; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > "$OUT/tests.log"; cat "$OUT/tests.log"
for s in 20 22; do timeout 900 python bench.py --workload mxm_plus_times_masked --scale $s --steps 3 --warmup 1 > "$OUT/mxm_masked_s$s.json" 2> "$OUT/mxm_masked_s$s.err"; echo "rc=$?"; cut -c1-300 "$OUT/mxm_masked_s$s.json"; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o mm -- python bench.py --workload mxm_plus_times_masked --scale 20 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/prof.json" 2> "$OUT/prof.err"
cp "$OUT/prof/mm_kernel_stats.csv" "$OUT/mxm_masked_s20_kernel_stats.csv"
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
