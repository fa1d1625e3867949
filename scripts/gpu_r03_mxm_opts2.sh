#!/bin/bash
# round 3, after the unit-kernel changes: the run-time thresholds once more
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
one() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --workload mxm_plus_times --scale 20 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), 'verified', d['verified'])"
}
one default
one min_cnt256 GRB_MXM_BITMAP_MIN_CNT=256
one min_cnt1024 GRB_MXM_BITMAP_MIN_CNT=1024
one minflops512 GRB_MXM_UNIT_MIN_FLOPS=512
one minflops2048 GRB_MXM_UNIT_MIN_FLOPS=2048
one perwin8 GRB_MXM_UNIT_MIN_PER_WINDOW=8
one perwin32 GRB_MXM_UNIT_MIN_PER_WINDOW=32
one mid2048 GRB_MXM_UNIT_MID=2048
one dense8192 GRB_MXM_UNIT_DENSE=8192
one dense3072 GRB_MXM_UNIT_DENSE=3072
one symwin16 GRB_MXM_SYM_WINDOWS=16
one default
