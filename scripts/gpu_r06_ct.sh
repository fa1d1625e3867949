#!/bin/bash
OUT=gpurun_out/r06h; mkdir -p $OUT; export TMPDIR=/tmp
for v in ${VARIANTS:-default ctpipe}; do
  if [ $v = default ]; then unset GRB_MI355X_LIB; else export GRB_MI355X_LIB=$PWD/build/variants/$v/libgrb_mi355x.so; fi
  python bench.py --steps 30 --no-cpu-baseline --no-extra 2>$OUT/$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), 'verified', d['verified'])"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o b -- python bench.py --steps 10 --no-cpu-baseline --no-extra > /dev/null 2>&1
  python - $OUT/$v/b_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "grb::k_mxv" in r["Name"] and ("ctile" in r["Name"] or "hstrip" in r["Name"] or "rtile" in r["Name"]):
        print("   ", r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
find $OUT -name '*kernel_trace.csv' -size +1M -delete
