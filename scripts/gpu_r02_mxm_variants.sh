#!/bin/bash
# round 2: build-time variants of the SpGEMM unit kernels (build/variants/<DEFINE>/libgrb_mi355x.so), scale $1
mkdir -p gpurun_out/mxmvar
one() { python bench.py --workload mxm_plus_times --scale ${SC} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2))"; }
SC=${1:-20}
unset GRB_MI355X_LIB; one default
for lib in build/variants/*/libgrb_mi355x.so; do
  GRB_MI355X_LIB="$PWD/$lib" one "$(basename "$(dirname "$lib")")"
done
unset GRB_MI355X_LIB; one default
