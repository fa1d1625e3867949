#!/bin/bash
# Round 5: GrX_Matrix_hint_ranked -- a graph relabelled by popularity up front, rows dealt block-cyclically: what one rank of a sharded run computes
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_ranked; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified','layout_build_call_ms')}, round((d.get('roofline') or {}).get('frac') or 0, 4), {k: (d.get('stats') or {}).get(k) for k in ('ordered','reorders','fused_epilogue','long_kernel')})"; }
echo "== parity"; timeout 600 python -m pytest tests/test_vertex_order.py -m gpu -x -q -k "ranked_hint" 2>&1 | tail -2
H="python bench.py --no-cpu-baseline --no-extra --ranked"
timeout 400 $H > "$OUT/whole.json" 2>"$OUT/whole.err"; line "$OUT/whole.json" "s24 whole, ranked labels"
for b in 0/2 0/4 0/8 3/8 7/8; do f="$OUT/block_$(echo $b | tr / _).json"; timeout 400 $H --block $b > "$f" 2>"$f.err"; line "$f" "s24 ranked block $b"; done
timeout 600 $H --workload kron26 --block 0/8 > "$OUT/kron26_0_8.json" 2>"$OUT/kron26.err"; line "$OUT/kron26_0_8.json" "kron26 ranked block 0/8"
timeout 600 $H --workload kron26 --block 5/8 > "$OUT/kron26_5_8.json" 2>/dev/null; line "$OUT/kron26_5_8.json" "kron26 ranked block 5/8"
timeout 400 $H --workload mxv_lor_land_masked --block 0/8 > "$OUT/bfs_0_8.json" 2>/dev/null; line "$OUT/bfs_0_8.json" "BFS step ranked block 0/8"
timeout 400 python bench.py --no-cpu-baseline --no-extra --workload mxv_lor_land_masked --block 0/8 > "$OUT/bfs_0_8_natural.json" 2>/dev/null; line "$OUT/bfs_0_8_natural.json" "BFS step natural block 0/8"
tail -3 "$OUT/whole.err"
