#!/bin/bash
# Round 5: ranked labels, block-cyclic row dealing -- stripe height against balance (ranks 0 and 7 of 8), scale 24 and Kronecker 26
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_ranked2; mkdir -p "$OUT"; export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', {k: d.get(k) for k in ('ms_per_step','verified')}, round((d.get('roofline') or {}).get('frac') or 0, 4), int(d['config']['edges_counted_per_step']), (d.get('stats') or {}).get('ordered'))"; }
H="python bench.py --no-cpu-baseline --no-extra --ranked"
for B in 1 8 64 512; do for b in 0/8 7/8; do f="$OUT/s24_B${B}_$(echo $b | tr / _).json"; timeout 400 $H --stripe $B --block $b > "$f" 2>/dev/null; line "$f" "s24 stripe $B block $b"; done; done
for b in 0/8 7/8; do f="$OUT/kron26_B8_$(echo $b | tr / _).json"; timeout 600 $H --workload kron26 --stripe 8 --block $b > "$f" 2>/dev/null; line "$f" "kron26 stripe 8 block $b"; done
for b in 0/4 0/2; do f="$OUT/s24_B8_$(echo $b | tr / _).json"; timeout 400 $H --stripe 8 --block $b > "$f" 2>/dev/null; line "$f" "s24 stripe 8 block $b"; done
