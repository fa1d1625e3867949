#!/bin/bash
# host side of the cold calls: HIP API time (rocprofv3 --hip-trace --stats) of bench with 1 timed step
OUT=gpurun_out/firstapi; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d "$OUT/p" -o b -- python bench.py --steps 1 --warmup 2 --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
ls "$OUT/p"
python - "$OUT/p" <<'PY'
import csv, sys, glob, os
for f in glob.glob(os.path.join(sys.argv[1], "*hip_api_stats.csv")) + glob.glob(os.path.join(sys.argv[1], "*hip_stats.csv")):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:16]:
        print("   ", r["Name"][:50].ljust(50), r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), "ms total", round(float(r["MaxNs"]) / 1e6, 2), "ms max")
PY
python -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('first', d.get('first_call_ms'), 'second', d.get('layout_build_call_ms'))"
find "$OUT" -name '*_trace.csv' -size +1M -delete
