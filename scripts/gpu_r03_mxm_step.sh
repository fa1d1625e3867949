#!/bin/bash
# round 3: one SpGEMM measurement step -- bench line at scale 20 (+ scale 22 streamed with FULL=1), per-kernel stats, VALU / LDS instruction counts
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-step}
O=gpurun_out/mxm_$TAG; mkdir -p $O
timeout 600 python bench.py --workload mxm_plus_times --scale 20 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2> $O/s20.err > $O/s20.json
python -c "import json; d=json.load(open('$O/s20.json')); print('s20', round(d['ms_per_step'],2), 'ms verified', d['verified'], 'frac', round(d['roofline']['frac'],3))"
if [ -n "$FULL" ]; then
  timeout 900 python bench.py --workload mxm_plus_times --scale 22 --steps 2 --warmup 1 --no-cpu-baseline --no-extra 2> $O/s22.err > $O/s22.json
  python -c "import json; d=json.load(open('$O/s22.json')); print('s22 streamed', round(d['ms_per_step'],1), 'ms verified', d['verified'], 'frac', round(d['roofline']['frac'],3))"
  timeout 900 python bench.py --workload mxm_plus_times_masked --scale 22 --steps 2 --warmup 1 --no-cpu-baseline --no-extra 2> $O/s22m.err > $O/s22m.json
  python -c "import json; d=json.load(open('$O/s22m.json')); print('s22 masked', round(d['ms_per_step'],1), 'ms verified', d['verified'])"
fi
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o p -- python $OLDPWD/bench.py --workload mxm_plus_times --scale 20 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1)
python - "$(find $O/prof -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'grb::' in r['Name'] and float(r['TotalDurationNs']) > 1e6:
        print('   %-64s calls %3s per_pass_ms %8.2f' % (r['Name'][10:74], r['Calls'], float(r['TotalDurationNs']) / 1e6 / 3))
PY
if [ -z "$NOPMC" ]; then bash scripts/gpu_pmc_mxm.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" 2>&1 | tail -7; fi
