#!/bin/bash
# PMC counters of the SpGEMM kernels (scale 20):  gpu_pmc_mxm.sh "CTR CTR ..." ["CTR ..." ...]   (one rocprofv3 pass per group)
OUT=gpurun_out/pmc_mxm; mkdir -p "$OUT"; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1)); d="$OUT/g$i"
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -o c -- python bench.py --workload mxm_plus_times --scale 20 --steps 1 --warmup 0 --no-cpu-baseline > "$d.json" 2> "$d.err"
  echo "== [$grp] rc=$?"
  python - "$d" <<'PY'
import csv, sys, glob, os, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_spgemm_unit" in k:
            agg[k.split("(")[0][10:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print("  ", k.ljust(52), {c: f"{v:.4g}" for c, v in sorted(d.items())})
PY
  find "$d" -name '*counter_collection.csv' -size +2M -delete; find "$d" -name '*kernel_trace.csv' -size +1M -delete
done
