#!/bin/bash
# round 3: the wavefront write rule (k_mat_write_wave) -- parity on the GPU, then the unfused complemented mask (full product +
# write rule over 9.7 G entries) with the new kernel against the thread-per-row kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/write; mkdir -p $O
(time timeout 900 python -m pytest tests/test_random_parity.py tests/test_gpu_scale.py tests/test_reference_parity.py -x -q -m gpu -k "mxm or write_rule or transpose" -n 4) > $O/tests.log 2>&1
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
one cmask_s18_unfused_wave "--workload mxm_plus_times_cmask --scale 18 --steps 3 --warmup 1" GRB_MXM_MASK_MODE=0
one cmask_s18_unfused_thread "--workload mxm_plus_times_cmask --scale 18 --steps 3 --warmup 1" GRB_MXM_MASK_MODE=0 GRB_MAT_WRITE_KERNEL=0
one cmask_s19_unfused_wave "--workload mxm_plus_times_cmask --scale 19 --steps 3 --warmup 1" GRB_MXM_MASK_MODE=0
one cmask_s20_unfused_wave "--workload mxm_plus_times_cmask --scale 20 --steps 2 --warmup 1" GRB_MXM_MASK_MODE=0
cd /tmp && GRB_MXM_MASK_MODE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o p -- python /root/repo/bench.py --no-cpu-baseline --no-extra --workload mxm_plus_times_cmask --scale 19 --steps 2 --warmup 1 > /root/repo/$O/prof.json 2> /root/repo/$O/prof.err
cd /root/repo
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep "grb::" "$f" | head -8 | cut -c1-150
