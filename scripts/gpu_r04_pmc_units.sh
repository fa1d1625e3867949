#!/bin/bash
# Round 4: what are the mxv kernels busy with?  Derived rocprofv3 metrics per kernel of the headline call (separate passes).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_pmc_units; mkdir -p "$OUT"; export TMPDIR=/tmp
WL=${WL:-mxv_min_plus}
for c in ${CTRS:-"MemUnitBusy" "MemUnitStalled" "VALUBusy" "LDSBankConflict" "TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"}; do
  d="$OUT/pmc_$c"
  ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$d" -o p -- python $OLDPWD/bench.py --workload $WL --steps 2 --warmup 3 --no-cpu-baseline --no-extra > "$OLDPWD/$d.json" 2> "$OLDPWD/$d.err" ); echo "pmc [$c] rc=$?"
done
python - "$OUT" <<'PY'
import sys, glob, csv, os, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "grb::k_mxv" not in k and "k_long_init" not in k: continue
        acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v); print("    %-40s median %.4g  (n=%d)" % (c, v[len(v)//2], len(v)))
PY
