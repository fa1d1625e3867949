#!/bin/bash
# Round 4, last validation: the whole GPU tier, smoke(), the default bench line, then the thin-frontier points of the push / pull grid
# at scale 22 with the one-workgroup push (default) and without (GRB_PUSH_SMALL=0), and the BFS loop.
cd "$(dirname "$0")/.."
bash scripts/gpu_r04_full.sh
OUT=gpurun_out/r04_final2; mkdir -p "$OUT"
show() { python - "$1" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if not l.startswith("{"): continue
    d = json.loads(l)
    print(d["scale"], d.get("frontier_density", d.get("u_density")), d.get("visited_density"), {k: round(d[k], 4) for k in ("pull_ms", "push_ms", "auto_ms")}, d["auto_method"], d.get("work"))
PY
}
timeout 600 python scripts/push_vs_pull.py 22 > "$OUT/push_pull_grid_s22.jsonl" 2> "$OUT/grid.err"; echo "== push/pull grid scale 22 (push_small 1)"; show "$OUT/push_pull_grid_s22.jsonl"; tail -2 "$OUT/grid.err"
GRB_PUSH_SMALL=0 timeout 600 python scripts/push_vs_pull.py 22 > "$OUT/push_pull_grid_s22_push_small_0.jsonl" 2>> "$OUT/grid.err"; echo "== push/pull grid scale 22 (push_small 0)"; show "$OUT/push_pull_grid_s22_push_small_0.jsonl"
timeout 900 python bench.py --workload bfs --steps 5 --no-cpu-baseline > "$OUT/bfs_s24.json" 2> "$OUT/bfs.err"
python -c "
import json; d=json.loads(open('$OUT/bfs_s24.json').read().strip().splitlines()[-1]); print('bfs', d.get('value'), d.get('ms_per_step'), [(x.get('frontier'), x.get('direction'), x.get('ms')) for x in (d['config'].get('per_level') or [])])"
tail -2 "$OUT/bfs.err"
