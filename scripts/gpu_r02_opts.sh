#!/bin/bash
# quick option sweeps on the masked / unmasked min_plus call: short-row kernel, thresholds
one() {
  tag=$1; shift
  env "$@" python bench.py --steps 20 --no-cpu-baseline --extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), [(e['workload'], round(e['ms_per_step'],4)) for e in d['extra']])"
}
one default GRB_X=0
one sell_4096 GRB_SHORT_KERNEL=2
one sell_1024 GRB_SHORT_KERNEL=2 GRB_SELL_SIGMA=1024
one t48 GRB_SPLIT_MIN_LEN=48
one t96 GRB_SPLIT_MIN_LEN=96
one t32_sell GRB_SPLIT_MIN_LEN=32 GRB_SHORT_KERNEL=2
one nc32_t96 GRB_SPLIT_MIN_LEN=96 GRB_LONG_CLASSES=32
