#!/bin/bash
# Round 2: class strips (long_kernel 2) against class items (1) at several long-row thresholds; all three mxv workloads.
TAG=${1:-r02c}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== gpu tests (split paths)"; timeout 900 python -m pytest tests/test_random_parity.py -x -q -m gpu -k "split or long_rows or reductions_over or terminal or bool" 2>&1 | tail -3
one() {
  python bench.py --steps 20 --no-cpu-baseline --extra 2>"$OUT/err.$1" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), [(e['workload'], round(e['ms_per_step'],4)) for e in d['extra']], d['stats'])"
}
GRB_LONG_KERNEL=1 one "items_256"
for t in 256 128 64 32 16; do GRB_LONG_KERNEL=2 GRB_SPLIT_MIN_LEN=$t one "strip_$t"; done
