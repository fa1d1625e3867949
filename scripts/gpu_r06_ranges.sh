#!/bin/bash
# Round 6: the cut of the cold tiles' column ranges (GRB_ORD_RANGES = target count by equal reference counts, GRB_ORD_RANGE_KB = width cap in KiB of operand)
OUT=gpurun_out/r06w; mkdir -p $OUT
one() { tag=$1; shift; env "$@" GRB_PRINT_STRIPS=1 python bench.py --steps 30 --no-cpu-baseline --no-extra --no-warm-build 2>$OUT/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'])"; grep "cold tiles" $OUT/$tag.err | cut -c1-70; }
one default A=1
for r in 16 24 32 40 48; do for kb in 1024 2048 4096; do one r${r}_kb${kb} GRB_ORD_RANGES=$r GRB_ORD_RANGE_KB=$kb; done; done
one default_b A=1
