#!/bin/bash
# Round 4: the lines that run on the round-2 layouts (matrices below lean_min_nnz: scale 20 / 22, one rank's row block of an 8-way run)
# and the BOOL level step -- the kernels whose LDS heads are filled by lds_fill
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_small; mkdir -p "$OUT"
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['ms_per_step'],4), d.get('verified'), round(d['roofline']['frac'],3))"; }
timeout 300 python bench.py --scale 20 --no-extra --no-cpu-baseline > "$OUT/s20.json" 2>/dev/null; line "$OUT/s20.json" "scale 20 masked min_plus"
timeout 300 python bench.py --scale 22 --no-extra --no-cpu-baseline > "$OUT/s22.json" 2>/dev/null; line "$OUT/s22.json" "scale 22 masked min_plus"
timeout 300 python bench.py --workload mxv_lor_land_masked --no-extra --no-cpu-baseline > "$OUT/lor_land_s24.json" 2>/dev/null; line "$OUT/lor_land_s24.json" "scale 24 BFS level step"
timeout 300 python bench.py --block 0/8 --no-extra --no-cpu-baseline > "$OUT/block0of8.json" 2>/dev/null; line "$OUT/block0of8.json" "block 0/8 of scale 24"
timeout 300 python bench.py --no-extra --no-cpu-baseline > "$OUT/s24.json" 2>/dev/null; line "$OUT/s24.json" "scale 24 masked min_plus"
