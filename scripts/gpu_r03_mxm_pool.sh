#!/bin/bash
# round 3: which units keep their bitmap from the symbolic pass (mxm_bitmap_min_cnt) and how large the pool may grow
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/mxm_pool; mkdir -p $O
one() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --workload mxm_plus_times --scale 20 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],2), 'verified', d['verified'])"
}
one default
one min256 GRB_MXM_BITMAP_MIN_CNT=256
one min128 GRB_MXM_BITMAP_MIN_CNT=128
one min128_pool64g GRB_MXM_BITMAP_MIN_CNT=128 GRB_MXM_BITMAP_POOL_MB=65536
one min32_pool64g GRB_MXM_BITMAP_MIN_CNT=32 GRB_MXM_BITMAP_POOL_MB=65536
one min1024 GRB_MXM_BITMAP_MIN_CNT=1024
one small256 GRB_MXM_UNIT_SMALL=256
one mid2048 GRB_MXM_UNIT_MID=2048
one minflops4096 GRB_MXM_UNIT_MIN_FLOPS=4096
one perwin32 GRB_MXM_UNIT_MIN_PER_WINDOW=32
