#!/bin/bash
# PMC counters of the kernels of one bench workload:  gpu_pmc.sh TAG WORKLOAD "COUNTER COUNTER ..." [ENV=VAL ...]
TAG=$1; WL=$2; CTRS=$3; shift 3; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/p" -o c -- python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/b.json" 2> "$OUT/b.err"
echo "== $WL [$CTRS] $* rc=$?"
python - "$OUT" <<'PY'
import csv, sys, glob, os, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "p", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if any(t in k for t in ("grb::k_mxv", "k_strip_act")):
            agg[k.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("  ", k.ljust(48), {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
find "$OUT" -name '*counter_collection.csv' -size +2M -delete; find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
