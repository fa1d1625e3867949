#!/bin/bash
# Round 4: the thin-frontier points of the push / pull grid at scale 22, one-workgroup push on (default) and off
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_push3; mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if not l.startswith("{"): continue
    d = json.loads(l)
    print(d["scale"], d.get("frontier_density", d.get("u_density")), d.get("frontier_with_edges"), d.get("visited_density"), {k: round(d[k], 4) for k in ("pull_ms", "push_ms", "auto_ms")}, d["auto_method"], d.get("work"), d.get("one_workgroup"))
PY
}
timeout 600 python scripts/push_vs_pull.py 22 > "$OUT/push_pull_grid_s22.jsonl" 2> "$OUT/grid.err"; echo "== push_small 1"; show "$OUT/push_pull_grid_s22.jsonl"; tail -2 "$OUT/grid.err"
GRB_PUSH_SMALL=0 timeout 600 python scripts/push_vs_pull.py 22 > "$OUT/push_pull_grid_s22_push_small_0.jsonl" 2>> "$OUT/grid.err"; echo "== push_small 0"; show "$OUT/push_pull_grid_s22_push_small_0.jsonl"
