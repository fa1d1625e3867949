#!/bin/bash
# round 3, GPU session 13: write rule without the unchanged stores; BFS step long-row thresholds
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
one() {  # tag env...
  tag=$1; shift
  env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra ${BENCH_WL:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'long_kernel', d['stats']['long_kernel'])"
}
{
one default
BENCH_WL="--workload mxv_min_plus" one unmasked
export BENCH_WL="--workload mxv_lor_land_masked"
one bfs_default
one bfs_len128 GRB_SPLIT_MIN_LEN=128
one bfs_len64 GRB_SPLIT_MIN_LEN=64
one bfs_len512 GRB_SPLIT_MIN_LEN=512
} > $O/ab.txt 2>&1
cat $O/ab.txt
