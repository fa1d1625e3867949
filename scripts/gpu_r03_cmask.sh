#!/bin/bash
# round 3: C<!M> = A (+.x) B with the complemented mask fused into the product -- parity on the GPU, then timings fused / unfused
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/cmask; mkdir -p $O
(time timeout 900 python -m pytest tests/test_random_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "mxm" -n 4) > $O/tests.log 2>&1
tail -3 $O/tests.log
one() {  # tag args env...
  tag=$1; args=$2; shift 2
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra $args 2> $O/$tag.err > $O/$tag.json
  python - "$O/$tag.json" "$tag" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], round(d["ms_per_step"], 2), "ms  nnz(C)", int(d["config"]["nnz_C"]), "verified", d["verified"], "frac", round(d["roofline"]["frac"], 3), "method", d["stats"]["method"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
one cmask_s18_fused "--workload mxm_plus_times_cmask --scale 18 --steps 3 --warmup 1"
one cmask_s18_unfused "--workload mxm_plus_times_cmask --scale 18 --steps 3 --warmup 1" GRB_MXM_MASK_MODE=0
one plain_s18 "--workload mxm_plus_times --scale 18 --steps 3 --warmup 1"
one cmask_s20_fused "--workload mxm_plus_times_cmask --scale 20 --steps 3 --warmup 1"
one plain_s20 "--workload mxm_plus_times --scale 20 --steps 3 --warmup 1"
one cmask_s20_unfused "--workload mxm_plus_times_cmask --scale 20 --steps 2 --warmup 1" GRB_MXM_MASK_MODE=0
