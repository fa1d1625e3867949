#!/bin/bash
TAG=${1:-r01b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.log"; tail -15 "$OUT/pytest_gpu.log"
echo "== ablation"; timeout 1500 python scripts/ablate_pull.py 24 > "$OUT/ablate.jsonl" 2> "$OUT/ablate.err"; echo "ablate rc=$?" | tee -a "$OUT/summary.log"; cat "$OUT/ablate.jsonl"; tail -5 "$OUT/ablate.err"
echo "== pmc"; 
timeout 900 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc1" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc1.json" 2> "$OUT/pmc1.err"; echo "pmc1 rc=$?" | tee -a "$OUT/summary.log"
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc2" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc2.json" 2> "$OUT/pmc2.err"; echo "pmc2 rc=$?" | tee -a "$OUT/summary.log"
for d in pmc1 pmc2; do f=$(find "$OUT/$d" -name '*counter_collection.csv' | head -1); echo "$f"; python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")
    if "k_mxv" in k:
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
done
find "$OUT" -name '*kernel_trace.csv' -size +5M -delete; find "$OUT" -name '*counter_collection.csv' -size +20M -delete
