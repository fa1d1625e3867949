#!/bin/bash
# Round 5: the bottom-up probe of the long BOOL rows (k_long_init, option bool_probe): the BFS level step and the whole BFS loop at probe depths 0 / 4 / 8 / 16
set -u
cd /root/repo
out=gpurun_out/r05_probe; mkdir -p $out
[ -n "${SKIP_PYTEST:-}" ] || timeout 900 python -m pytest tests/test_vertex_order.py tests/test_random_parity.py -m gpu -x -q 2>&1 | tail -3
for k in ${DEPTHS:-0 4 8 16}; do
  GRB_BOOL_PROBE=$k timeout 600 python bench.py --workload mxv_lor_land_masked --no-cpu-baseline --no-extra > $out/step_k$k.json 2> $out/step_k$k.err
  GRB_BOOL_PROBE=$k timeout 600 python bench.py --workload bfs --steps 5 --no-cpu-baseline > $out/bfs_k$k.json 2> $out/bfs_k$k.err
  python - <<PY
import json
for f in ("step","bfs"):
    try:
        x=json.loads(open("$out/%s_k$k.json"%f).read().strip().splitlines()[-1]); print("probe $k", f, round(x["ms_per_step"],4), x["verified"], round((x.get("roofline") or {}).get("frac",0),4), (x.get("stats") or {}).get("long_probe"))
    except Exception as e: print("probe $k", f, "failed", e)
PY
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_probe -o b -- python /root/repo/bench.py --workload mxv_lor_land_masked --no-cpu-baseline --no-extra > /dev/null 2>&1 ); cp $(find /tmp/prof_probe -name '*kernel_stats.csv' | head -1) $out/step_kernel_stats.csv
grep -E "grb::k_(mxv|long|pack|rows)" $out/step_kernel_stats.csv | cut -c1-130 | head -10
