#!/bin/bash
# Round 5, last session, after the LDS change of the SpGEMM unit kernels: the whole GPU test tier, the default bench command (headline + extras +
# cpu_baseline: the lines the driver reads), rocprofv3 kernel stats of the scale-22 product, the mask-driven product at scale 20.
cd "$(dirname "$0")/.."
TAG=${1:-r05final2}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value','unit','ms_per_step','verified')}, (d.get('roofline') or {}).get('frac')); [print('   extra:', (e.get('metric') or e.get('config',{}).get('workload',''))[:70], round(e.get('ms_per_step',0),4), e.get('verified'), (e.get('roofline') or {}).get('frac')) for e in d.get('extra',[])]"; }
echo "== pytest -m gpu"; ( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > "$OUT/tests_gpu_full.log" 2>&1; tail -6 "$OUT/tests_gpu_full.log"
echo "== bench default (headline + extras)"; ( time timeout 900 python bench.py ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; line "$OUT/bench_default.json"
echo "== rocprofv3 stats of the scale-22 product"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_${TAG}_mxm22" -o bench -- python $OLDPWD/bench.py --workload mxm_plus_times --scale 22 --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OLDPWD/$OUT/mxm_s22_under_rocprof.json 2>/dev/null ); cp $(find /tmp/prof_${TAG}_mxm22 -name '*kernel_stats.csv' | head -1) $OUT/mxm_s22_kernel_stats.csv; grep -E "grb::k_spgemm" "$OUT/mxm_s22_kernel_stats.csv" | cut -c1-140 | head -8
echo "== masked SpGEMM scale 20"; timeout 300 python bench.py --workload mxm_plus_times_masked --scale 20 --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/mxm_masked_s20.json" 2>/dev/null; line $OUT/mxm_masked_s20.json
