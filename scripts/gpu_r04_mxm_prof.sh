#!/bin/bash
# Round 4: rocprofv3 kernel stats (whole CSV) of the SpGEMM lines: scale 20 materialised, scale 22 streamed
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_mxm; mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$PWD
for cfg in "20 " "22 --streamed"; do
  set -- $cfg; sc=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_mxm_$sc" -o mxm -- python $ROOT/bench.py --workload mxm_plus_times --scale $sc --steps 2 --warmup 1 --no-cpu-baseline "$@" > $ROOT/$OUT/mxm_s$sc.json 2> $ROOT/$OUT/mxm_s$sc.err )
  cp $(find /tmp/prof_mxm_$sc -name '*kernel_stats.csv' | head -1) $OUT/mxm_s${sc}_kernel_stats.csv
  python -c "import json;d=json.loads(open('$OUT/mxm_s$sc.json').read().strip().splitlines()[-1]);print('scale',$sc,d['ms_per_step'],d['verified'],d['roofline']['frac'],d['config'].get('batches'))"
  python - "$OUT/mxm_s${sc}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("   total kernel time %.1f ms over 3 products" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("   %-90s calls %6s total %9.2f ms avg %9.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
done
