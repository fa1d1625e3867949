#!/bin/bash
# round 3: which kernels the layout-building second call of the headline matrix spends its time in
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/build; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_build -o b -- python $OLDPWD/bench.py --steps 3 --no-extra --no-cpu-baseline > /dev/null 2>&1 )
f=$(find /tmp/prof_build -name '*kernel_stats.csv' | head -1)
cp "$f" $O/kernel_stats_build.csv
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) <= 4 and "at::" not in r["Name"] and "elementwise" not in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = 0
for r in rows[:28]:
    print(r["Name"][:90].ljust(90), r["Calls"], round(float(r["TotalDurationNs"]) / 1e3, 1), "us")
PY
