#!/bin/bash
# usage: gpu_round3.sh TAG [parts]   parts: t=pytest a=ablation b=bench p=kernel stats
TAG=${1:-r01c}; PARTS=${2:-tab}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
if [[ $PARTS == *t* ]]; then echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.log"; tail -15 "$OUT/pytest_gpu.log"; fi
if [[ $PARTS == *a* ]]; then echo "== ablation"; timeout 900 python scripts/ablate_pull.py 24 ${ABL_IPT:-0,4} ${ABL_DBG:-0,1,2,7} > "$OUT/ablate.jsonl" 2> "$OUT/ablate.err"; echo "ablate rc=$?" | tee -a "$OUT/summary.log"; cat "$OUT/ablate.jsonl"; tail -3 "$OUT/ablate.err"; fi
if [[ $PARTS == *b* ]]; then echo "== bench"; timeout 900 python bench.py --steps 20 --extra > "$OUT/bench_s24.json" 2> "$OUT/bench_s24.err"; echo "bench rc=$?" | tee -a "$OUT/summary.log"; cat "$OUT/bench_s24.json"; tail -3 "$OUT/bench_s24.err"; fi
if [[ $PARTS == *p* ]]; then echo "== rocprofv3 stats"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 20 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err"; echo "rocprof rc=$?" | tee -a "$OUT/summary.log"; grep -E "k_mxv|k_spgemm" "$OUT/prof/bench_kernel_stats.csv" | cut -c1-200; find "$OUT/prof" -name '*kernel_trace.csv' -size +5M -delete; fi
