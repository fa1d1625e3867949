#!/bin/bash
# first / second call of the headline bench with and without the code objects loaded at GrB_init
for p in 1 0 1; do
  GRB_PRELOAD=$p python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('preload=$p first', round(d['first_call_ms'],2), 'build', round(d['layout_build_call_ms'],2), 'steady', round(d['ms_per_step'],3))"
done
python - <<'PY'
import time, ctypes
t=time.perf_counter()
from tests.backend import bind
gb=bind("gpu")
print("import + GrB_init s:", round(time.perf_counter()-t,2))
PY
GRB_PRELOAD=0 python - <<'PY'
import time, ctypes
t=time.perf_counter()
from tests.backend import bind
gb=bind("gpu")
print("import + GrB_init s (no preload):", round(time.perf_counter()-t,2))
PY
