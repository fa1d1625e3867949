#!/bin/bash
# round 3: the small-matrix path (round-2 layouts, k_mxv_rows / k_mxv_pull) after the DPP scans, and the write rule with readlane
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/small2; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 -k "not mxm_bench_sizes and not kron26" 2>&1 | tail -2
one() {
  tag=$1; args=$2
  python bench.py --steps 50 --no-cpu-baseline --no-extra $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'frac', round(d['roofline']['frac'],3))"
}
one "s20 masked (configs[1])" "--scale 20"
one "s20 unmasked" "--scale 20 --workload mxv_min_plus"
one "s22 masked" "--scale 22"
one "block 0/8" "--block 0/8"
one "block 0/16" "--block 0/16"
one "config 0 (uniform fp64)" "--workload uniform_fp64" 
one "s24 headline" ""
GRB_MXM_MASK_MODE=0 timeout 600 python bench.py --no-cpu-baseline --no-extra --workload mxm_plus_times_cmask --scale 19 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cmask s19 unfused (write rule)', round(d['ms_per_step'],2), d['verified'])"
