#!/bin/bash
# the round-end checks: the whole GPU tier, smoke(), the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/full; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/tests_gpu.log 2>&1
tail -4 $O/tests_gpu.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/full/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','verified','first_call_ms','layout_build_call_ms','preprocess_bytes')}, d['roofline']['frac'])
for e in d.get('extra',[]):
    print(e['workload'][:46], e.get('value'), e.get('ms_per_step'), e.get('verified'), (e.get('roofline') or {}).get('frac'), e.get('error'))
PY
