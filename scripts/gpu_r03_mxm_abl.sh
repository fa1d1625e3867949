#!/bin/bash
# round 3: what bounds the symbolic (row, window) units -- timing switches of the -DGRB_ABLATE build (results wrong on purpose)
#   32 = no search (products taken from the lane's own range of B), 64 = no bitmap atomics in the pattern pass
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/mxm_abl; mkdir -p $O
export GRB_MI355X_LIB=$PWD/build/variants/ABLATE/libgrb_mi355x.so
for bits in ${ABL_BITS:-0 128 256 320}; do
  flags=$((bits << 20))
  (cd /tmp && GRB_DEBUG_FLAGS=$flags rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/p$bits -o p -- python $OLDPWD/bench.py --workload mxm_plus_times --scale 20 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OLDPWD/$O/b$bits.json 2> $OLDPWD/$O/b$bits.err)
  f=$(find $O/p$bits -name "*kernel_stats.csv" | head -1)
  echo "== abl bits $bits"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_spgemm_unit' in r['Name']:
        print('   %-60s calls %3s avg_us %9.1f total_ms %8.2f' % (r['Name'][10:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done
