#!/bin/bash
# Round 4: the whole GPU tier (pytest -m gpu), smoke(), the default bench line
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_full; mkdir -p "$OUT"
( time timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > "$OUT/tests_gpu_full.log" 2>&1; tail -8 "$OUT/tests_gpu_full.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1200 python bench.py ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -4 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "verified", "first_call_ms", "layout_build_call_ms", "preprocess_bytes")}, d["roofline"]["frac"], d["cpu_baseline"])
for e in d.get("extra", []):
    print("  extra:", e.get("workload", "")[:70], {k: e.get(k) for k in ("ms_per_step", "verified", "error")}, (e.get("roofline") or {}).get("frac"))
PY
