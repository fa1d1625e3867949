#!/bin/bash
# round 3: the unfused path (full product + write rule) beside the fused complemented mask, and the kernel stats of the fused call
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/cmask; mkdir -p $O
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
one cmask_s18_unfused "--workload mxm_plus_times_cmask --scale 18 --steps 3 --warmup 1" GRB_MXM_MASK_MODE=0
one cmask_s19_fused "--workload mxm_plus_times_cmask --scale 19 --steps 3 --warmup 1"
one cmask_s19_unfused "--workload mxm_plus_times_cmask --scale 19 --steps 3 --warmup 1" GRB_MXM_MASK_MODE=0
one cmask_s20_unfused "--workload mxm_plus_times_cmask --scale 20 --steps 2 --warmup 1" GRB_MXM_MASK_MODE=0
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_fused -o p -- python /root/repo/bench.py --no-cpu-baseline --no-extra --workload mxm_plus_times_cmask --scale 20 --steps 2 --warmup 1 > /root/repo/$O/prof_fused.json 2> /root/repo/$O/prof_fused.err
cd /root/repo
f=$(find $O/prof_fused -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-170
