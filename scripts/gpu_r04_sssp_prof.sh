#!/bin/bash
# Round 4: rocprofv3 kernel stats of the SSSP loop (which kernels its sweeps run)
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sssp -o s -- python $GRAFT_REPO_ROOT/bench.py --workload sssp --steps 1 --warmup 1 --no-cpu-baseline > /tmp/s.json 2>/tmp/s.err
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_sssp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "grb::" in r["Name"])
print("grb kernel time total %.1f ms (3 solves of 11 sweeps)" % (tot/1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    if "grb::" in r["Name"]: print("  %-80s calls %4s avg %8.1f us total %7.2f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
