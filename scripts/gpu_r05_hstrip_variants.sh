#!/bin/bash
# Round 5: build variants of k_mxv_hstrip (scripts/build_variant.sh hs_<name> -D...) on the headline: step time, verified, the kernel's own time under rocprofv3
set -u
cd /root/repo
out=gpurun_out/r05_hs; mkdir -p $out
for v in ${VARIANTS:-default hs_depth4 hs_depth5 hs_abl2 hs_abl4}; do
  lib=""; [ "$v" != default ] && lib="$PWD/build/variants/$v/libgrb_mi355x.so"
  [ -n "$lib" ] && [ ! -f "$lib" ] && { echo "$v: no library"; continue; }
  ( cd /tmp && GRB_MI355X_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o b -- python /root/repo/bench.py --no-cpu-baseline --no-extra ${BENCH_ARGS:-} > /root/repo/$out/$v.json 2> /root/repo/$out/$v.err )
  python - <<PY
import json,csv,glob
try:
    x=json.loads(open("$out/$v.json").read().strip().splitlines()[-1]); print("$v", round(x["ms_per_step"],4), x["verified"], end="  ")
except Exception as e: print("$v failed", e, end="  ")
for f in glob.glob("/tmp/prof_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_mxv_hstrip","k_mxv_rtile","k_mxv_ctile")) and int(r["Calls"])>=20: print(r["Name"].split("<")[0].split("::")[-1], round(float(r["AverageNs"])/1e3,1), end="  ")
print()
PY
done
