#!/bin/bash
# Round 5: what ONE rank of the row-sharded scale-22 SpGEMM computes (flop-balanced share r of 8, B replicated), alone on one GPU -- the basis of
# profiles/r05/projected_scaling.md (a projection, no multi-GPU measurement)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_mxm_blocks; mkdir -p "$OUT"; export TMPDIR=/tmp
for b in 0/8 3/8 7/8 0/4 0/2; do
  f="$OUT/mxm_s22_block_$(echo $b | tr / _).json"
  timeout 600 python bench.py --workload mxm_plus_times --scale 22 --block $b --steps 2 --warmup 1 --no-cpu-baseline > "$f" 2>/dev/null
  python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$b', d['ms_per_step'], d['verified'], d['config']['flops'], d['config']['nnz_C'], d['config'].get('batches'))"
done
