#!/bin/bash
# per-kernel times (rocprofv3 stats) of one bench workload under the given environment:  gpu_kt.sh TAG WORKLOAD [ENV=VAL ...]
TAG=$1; WL=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p" -o b -- python bench.py --workload $WL --steps 10 --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
echo "== $WL $* rc=$? $(python -c "import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
python - "$OUT/p/b_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(t in r["Name"] for t in ("grb::k_mxv", "k_long_", "k_strip_act", "grb::k_x_image", "k_pack", "k_vec_write")):
        print("   ", r["Name"][:64].ljust(64), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
