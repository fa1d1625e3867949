#!/bin/bash
# round 3: the SpGEMM lines, per-kernel stats and counters after the unit-kernel changes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/mxm_final; mkdir -p $O
timeout 900 python bench.py --workload mxm_plus_times --scale 20 --steps 3 --warmup 1 --no-extra > $O/mxm_s20.json 2> $O/mxm_s20.err
timeout 900 python bench.py --workload mxm_plus_times --scale 22 --steps 2 --warmup 1 --no-extra > $O/mxm_s22_streamed.json 2> $O/mxm_s22.err
timeout 900 python bench.py --workload mxm_plus_times_cmask --scale 20 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/mxm_cmask_s20.json 2> $O/mxm_cmask_s20.err
python - <<'PY'
import json
for f in ("mxm_s20", "mxm_s22_streamed", "mxm_cmask_s20"):
    d = json.load(open(f"gpurun_out/mxm_final/{f}.json"))
    print(f, round(d["ms_per_step"], 2), "ms", "verified", d["verified"], "frac", round(d["roofline"]["frac"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o p -- python $OLDPWD/bench.py --workload mxm_plus_times --scale 20 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1)
cp "$(find $O/prof -name '*kernel_stats.csv' | head -1)" $O/mxm_s20_kernel_stats.csv
bash scripts/gpu_pmc_mxm.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" > $O/pmc_mxm_units_r03.txt 2>&1
cat $O/pmc_mxm_units_r03.txt | cut -c1-230
