#!/bin/bash
# Round 5, last session: the library changed outside the kernels (table copies staged through page-locked blocks, GRB_TRACE_COPIES), so the
# lines the driver reads are re-taken from the final library: the whole GPU test tier, the default bench command (headline + extras +
# cpu_baseline), rocprofv3 kernel stats of the default command and of the BFS level step.  (PMC passes and the other bench lines of
# scripts/gpu_final_r05.sh are kernel measurements of unchanged kernels: not repeated.)
cd "$(dirname "$0")/.."
TAG=${1:-r05final}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value','unit','ms_per_step','verified','layout_build_call_ms')}, (d.get('roofline') or {}).get('frac')); [print('   extra:', (e.get('metric') or e.get('config',{}).get('workload',''))[:70], round(e.get('ms_per_step',0),4), e.get('verified'), (e.get('roofline') or {}).get('frac')) for e in d.get('extra',[])]"; }
echo "== pytest -m gpu"; ( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > "$OUT/tests_gpu_full.log" 2>&1; tail -6 "$OUT/tests_gpu_full.log"
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench default (headline + extras)"; ( time timeout 900 python bench.py ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; line "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err"
echo "== rocprofv3 stats of the default command (headline workload only)"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$TAG" -o bench -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra > $OLDPWD/$OUT/bench_default_under_rocprof.json 2> $OLDPWD/$OUT/prof.err ); cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv; grep -E "grb::k_mxv|grb::k_long_init" "$OUT/bench_default_kernel_stats.csv" | cut -c1-150 | head -6
echo "== rocprofv3 stats of the BFS level step"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_${TAG}_bfs" -o bench -- python $OLDPWD/bench.py --workload mxv_lor_land_masked --no-cpu-baseline --no-extra > $OLDPWD/$OUT/bench_lor_land_under_rocprof.json 2>/dev/null ); cp $(find /tmp/prof_${TAG}_bfs -name '*kernel_stats.csv' | head -1) $OUT/bench_lor_land_kernel_stats.csv; grep -E "grb::k_" "$OUT/bench_lor_land_kernel_stats.csv" | cut -c1-120 | head -8
