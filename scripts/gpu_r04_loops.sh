#!/bin/bash
# Round 4: the traversal loops (BFS, SSSP) at scale 24 and the push / pull grid
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_loops; mkdir -p "$OUT"
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('value','ms_per_step')}, d['config'].get('ms_per_iteration'), d['config'].get('levels'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value')); print([ (x.get('frontier'), x.get('direction')) for x in (d['config'].get('per_level') or d['config'].get('per_sweep') or [])])"; }
timeout 900 python bench.py --workload bfs --steps 5 > "$OUT/bfs_s24.json" 2> "$OUT/bfs.err"; line "$OUT/bfs_s24.json" bfs; tail -2 "$OUT/bfs.err"
timeout 900 python bench.py --workload sssp --steps 3 > "$OUT/sssp_s24.json" 2> "$OUT/sssp.err"; line "$OUT/sssp_s24.json" sssp; tail -2 "$OUT/sssp.err"
GRB_ORDER_MODE=0 timeout 900 python bench.py --workload bfs --steps 5 --no-cpu-baseline > "$OUT/bfs_s24_order0.json" 2>/dev/null; line "$OUT/bfs_s24_order0.json" "bfs order0"
GRB_ORDER_MODE=0 timeout 900 python bench.py --workload sssp --steps 3 --no-cpu-baseline > "$OUT/sssp_s24_order0.json" 2>/dev/null; line "$OUT/sssp_s24_order0.json" "sssp order0"
timeout 1200 python scripts/push_vs_pull.py 24 > "$OUT/push_pull_grid.jsonl" 2> "$OUT/grid.err"; python - "$OUT/push_pull_grid.jsonl" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["scale"], d.get("frontier_density", d.get("u_density")), d.get("visited_density"), {k: round(d[k], 4) for k in ("pull_ms", "push_ms", "auto_ms")}, d["auto_method"], d.get("work"))
PY
