#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench lines, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== rocm-smi" > "$OUT/env.log"; (rocm-smi --showproductname 2>&1 | head -20; nproc; free -g | head -2) >> "$OUT/env.log"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/summary.log"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.log"; tail -5 "$OUT/pytest_gpu.log"
echo "== bench scale 20"; timeout 600 python bench.py --scale 20 --steps 20 > "$OUT/bench_s20.json" 2> "$OUT/bench_s20.err"; echo "bench20 rc=$?" | tee -a "$OUT/summary.log"; cat "$OUT/bench_s20.json"
echo "== bench scale 24"; timeout 1200 python bench.py --steps 20 --extra > "$OUT/bench_s24.json" 2> "$OUT/bench_s24.err"; echo "bench24 rc=$?" | tee -a "$OUT/summary.log"; cat "$OUT/bench_s24.json"
echo "== rocprofv3 kernel stats (scale 24)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 20 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rocprof rc=$?" | tee -a "$OUT/summary.log"
find "$OUT/prof" -name '*stats*' | head; for f in $(find "$OUT/prof" -name '*kernel_stats.csv'); do head -12 "$f"; done
# kernel traces are large: keep only the stats
find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete
tail -3 "$OUT"/*.err 2>/dev/null | tail -30
