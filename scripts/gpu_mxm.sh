#!/bin/bash
TAG=${1:-r01m}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q -k mxm > "$OUT/pytest_mxm.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_mxm.log"
for s in ${SCALES:-14 16 18}; do
  echo "== mxm scale $s"; timeout 900 python bench.py --workload mxm_plus_times --scale $s --steps 3 --warmup 1 > "$OUT/mxm_s$s.json" 2> "$OUT/mxm_s$s.err"; echo "rc=$?"; cut -c1-900 "$OUT/mxm_s$s.json"; tail -2 "$OUT/mxm_s$s.err"
done
