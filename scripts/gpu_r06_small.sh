#!/bin/bash
# Round 6: where the time of the scale-20 / 21 / 22 min_plus mxv (configs[1]: scale 20) goes -- kernel stats per scale
OUT=gpurun_out/r06k; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$VARIANT" ] && export GRB_MI355X_LIB=$PWD/build/variants/$VARIANT/libgrb_mi355x.so
for s in 20 21 22; do
  python bench.py --scale $s --workload mxv_min_plus --steps 50 --no-cpu-baseline --no-extra 2>$OUT/s$s.err | tee $OUT/s$s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scale $s', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'verified', d['verified'], 'launches', d['stats'].get('kernel_launches'), 'ordered', d['stats'].get('ordered'), 'long_kernel', d['stats'].get('long_kernel'))"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s$s -o b -- python $OLDPWD/bench.py --scale $s --workload mxv_min_plus --steps 50 --no-cpu-baseline --no-extra > /dev/null 2>&1 )
  cp /tmp/prof_s$s/*/b_kernel_stats.csv $OUT/s${s}_kernel_stats.csv 2>/dev/null || cp /tmp/prof_s$s/b_kernel_stats.csv $OUT/s${s}_kernel_stats.csv
  python - $OUT/s${s}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 50 and "grb::" in r["Name"]:
        print("   ", r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
