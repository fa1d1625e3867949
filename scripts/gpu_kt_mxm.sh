#!/bin/bash
# per-kernel times (rocprofv3 stats) of the SpGEMM bench:  gpu_kt_mxm.sh TAG SCALE [ENV=VAL ...]   (WL=mxm_plus_times_masked for the masked product)
TAG=$1; SC=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
env "$@" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p" -o b -- python bench.py --workload ${WL:-mxm_plus_times} --scale $SC --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
echo "== ${WL:-mxm_plus_times} scale $SC $* rc=$? $(python -c "import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
python - "$OUT/p/b_kernel_stats.csv" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("   ", r["Name"][:90].ljust(90), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
