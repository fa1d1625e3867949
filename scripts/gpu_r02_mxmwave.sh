#!/bin/bash
# round 2: (row, window) unit SpGEMM -- class limits / thresholds.  usage: gpu_r02_mxmwave.sh SCALE "ENV=VAL ENV=VAL" ...
mkdir -p gpurun_out/mxmwave; SC=$1; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python bench.py --workload mxm_plus_times --scale $SC --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/mxmwave/c$i.json 2> gpurun_out/mxmwave/c$i.err
  echo "$cfg: $(python -c "import json; d=json.loads(open('gpurun_out/mxmwave/c$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
done
