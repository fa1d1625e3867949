#!/bin/bash
# what the first GrB_mxv on a matrix costs: kernel times of a cold call (bench with 1 step, no warm-up)
TAG=${1:-first}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p" -o b -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
python - "$OUT/p/b_kernel_stats.csv" "$OUT/b.json" <<'PY'
import csv, sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("first_call_ms", d.get("first_call_ms"), "layout_build_call_ms", d.get("layout_build_call_ms"), "preprocess_bytes", d.get("preprocess_bytes"), "matrix_bytes", d.get("matrix_bytes"), "ms_per_step", d["ms_per_step"])
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "grb::" in r["Name"] or "rocprim" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = 0
for r in rows[:24]:
    print("   ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["TotalDurationNs"]) / 1e3, 1), "us total")
print("sum of library kernels (ms):", round(sum(float(r["TotalDurationNs"]) for r in rows) / 1e6, 2))
PY
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
