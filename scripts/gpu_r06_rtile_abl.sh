#!/bin/bash
# Round 6: ablation builds of k_mxv_rtile on the headline (results of the ablated builds are WRONG on purpose): per-kernel times under rocprofv3.
#   rt_fold      every gather folded into the first 2 MiB of the operand (all L2 hits): what cache residency of the operand would buy
#   rt_head*     gathers of codes below N not issued: what an LDS head of N codes would take away
#   rt_nogather  no operand gathers at all
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
python scripts/ref_shares.py --scale 24 > $OUT/ref_shares_s24.json 2> $OUT/ref_shares.err
for v in default rt_fold rt_head10k rt_head40k rt_head160k rt_nogather; do
  if [ $v = default ]; then unset GRB_MI355X_LIB; else export GRB_MI355X_LIB=$PWD/build/variants/$v/libgrb_mi355x.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o b -- python bench.py --steps 10 --no-cpu-baseline --no-extra > $OUT/$v.json 2> $OUT/$v.err
  echo "== $v $(python -c "import json; d=json.loads(open('$OUT/$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'])")"
  python - $OUT/$v/b_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "grb::k_mxv" in r["Name"] or "k_long_" in r["Name"]:
        print("   ", r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done > $OUT/summary.txt 2>&1
find $OUT -name '*kernel_trace.csv' -size +1M -delete
cat $OUT/summary.txt
