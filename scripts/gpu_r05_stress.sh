#!/bin/bash
# Round 5, last session: (1) the seeded parity tests with seeds beyond the ones pytest runs, on the GPU -- the round-5 paths (ordered layouts,
# sorted row tiles, bottom-up probe, ranked hint: STRESS_SET=order) and the random parity set; (2) a kernel + HIP-API timeline of the
# layout-building call of the headline matrix (scripts/build_timeline.py): where its 44 ms of wall time go.
#   PARTS="order random timeline" (default: all three)
cd "$(dirname "$0")/.."
ROOT=$PWD; O=gpurun_out/${TAG:-r05b}; mkdir -p $O; export TMPDIR=/tmp
PARTS=${PARTS:-order random timeline}
if [[ " $PARTS " == *" order "* ]]; then
  ( time STRESS_SET=order timeout 420 python scripts/stress_parity.py gpu 100 ${STRESS_ORDER_HI:-260} ) > $O/stress_order.txt 2>&1; tail -4 $O/stress_order.txt
fi
if [[ " $PARTS " == *" random "* ]]; then
  ( time STRESS_SET=random timeout 420 python scripts/stress_parity.py gpu 200 ${STRESS_RANDOM_HI:-260} ) > $O/stress_random.txt 2>&1; grep -v "^  File\|^    " $O/stress_random.txt | tail -12
fi
if [[ " $PARTS " == *" timeline "* ]]; then
  rm -rf /tmp/tl
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl -o tl -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $ROOT/$O/timeline_bench.json 2> $ROOT/$O/timeline.err )
  python scripts/build_timeline.py /tmp/tl > $O/layout_build_timeline.txt 2>&1; cat $O/layout_build_timeline.txt
fi
