#!/bin/bash
# round 3: symbolic units walking several windows of a row (rows of up to 128 entries of A), the long rows of A from a list
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/mxm_symwin; mkdir -p $O
for wg in 1 2 4 8 16 64; do
  (cd /tmp && GRB_MXM_SYM_WINDOWS=$wg rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/p$wg -o p -- python $OLDPWD/bench.py --workload mxm_plus_times --scale 20 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OLDPWD/$O/b$wg.json 2> $OLDPWD/$O/b$wg.err)
  python - "$O/b$wg.json" "$(find $O/p$wg -name '*kernel_stats.csv' | head -1)" $wg <<'PY'
import csv, sys, json
d = json.load(open(sys.argv[1]))
sym = sum(float(r['TotalDurationNs']) for r in csv.DictReader(open(sys.argv[2])) if 'k_spgemm_unit<long, 0' in r['Name']) / 1e6 / 3
print('windows per symbolic unit', sys.argv[3], ': product', round(d['ms_per_step'], 2), 'ms (under rocprof), symbolic units', round(sym, 2), 'ms per pass, verified', d['verified'])
PY
done
