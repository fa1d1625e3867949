#!/bin/bash
# Round 4: where the layout-building call goes (kernels that run once per matrix), and its wall time un-profiled
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_build; mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_build -o b -- python $ROOT/bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 3 > /dev/null 2> $ROOT/$OUT/err.txt )
cp $(find /tmp/prof_build -name '*kernel_stats.csv' | head -1) $OUT/layout_build_kernel_stats.csv
python - "$OUT/layout_build_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) <= 12 and float(r["TotalDurationNs"]) > 50e3 and "at::" not in r["Name"] and "k_mxv_pull" not in r["Name"] and "seams" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels of the layout build (calls <= 12): %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print("   %-80s calls %3s total %7.2f ms" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-extra --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('layout_build_call_ms', round(d['layout_build_call_ms'],1), 'first_call_ms', round(d['first_call_ms'],2), 'preprocess_bytes', d['preprocess_bytes'], 'matrix_bytes', d['matrix_bytes'])"; done
