#!/bin/bash
# per-kernel times of the scale-20 unmasked min_plus mxv (configs[1]) and of rank 0's block of an 8-way scale-24 run
export TMPDIR=/tmp
for cfg in "--scale 20 --workload mxv_min_plus" "--block 0/8"; do
  OUT=gpurun_out/kt_small; rm -rf "$OUT"; mkdir -p "$OUT"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p" -o b -- python bench.py $cfg --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
  echo "== $cfg: $(python -c "import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  python - "$OUT/p/b_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "grb::" in r["Name"] and int(r["Calls"]) >= 50]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("   ", r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
