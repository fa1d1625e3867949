#!/bin/bash
# round 3: small matrices (one rank's block of an 8- / 16-way run, scale 20): round-2 layouts against round-3 layouts
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/small; mkdir -p $O
one() {  # tag args -- env...
  tag=$1; args=$2; shift 2
  env "$@" python bench.py --steps 50 --no-cpu-baseline --no-extra $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'])"
}
{
for cfg in "--block 0/8" "--block 0/16" "--scale 20 --workload mxv_min_plus" "--scale 22 --workload mxv_min_plus" "--scale 22"; do
  one "r3 [$cfg]" "$cfg"
  one "r2-layouts [$cfg]" "$cfg" GRB_LONG_KERNEL=3 GRB_SHORT_KERNEL=1
  one "r2-long+tagged [$cfg]" "$cfg" GRB_LONG_KERNEL=3 GRB_SHORT_KERNEL=5
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
