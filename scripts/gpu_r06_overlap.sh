#!/bin/bash
# Round 6: the cold tiles on the auxiliary stream next to the hot strips (GRB_MXV_OVERLAP), and the hot strips on fewer workgroups (GRB_STRIP_WGS)
OUT=gpurun_out/r06d; mkdir -p $OUT; export TMPDIR=/tmp
one() { python bench.py --steps 30 --no-cpu-baseline --no-extra "${@:2}" 2>$OUT/$1.err | tee $OUT/$1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), 'verified', d['verified'], 'frac', round(d['roofline']['frac'],4))"; }
GRB_MXV_OVERLAP=0 one overlap0
GRB_MXV_OVERLAP=1 one overlap1
GRB_MXV_OVERLAP=1 GRB_STRIP_WGS=192 one overlap1_wgs192
GRB_MXV_OVERLAP=1 GRB_STRIP_WGS=128 one overlap1_wgs128
GRB_MXV_OVERLAP=0 GRB_STRIP_WGS=128 one overlap0_wgs128
GRB_MXV_OVERLAP=1 one overlap1_unmasked --workload mxv_min_plus
GRB_MXV_OVERLAP=0 one overlap0_unmasked --workload mxv_min_plus
GRB_MXV_OVERLAP=1 one overlap1_bfs --workload mxv_lor_land_masked
for v in 0 1; do
  GRB_MXV_OVERLAP=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_overlap$v -o b -- python bench.py --steps 10 --no-cpu-baseline --no-extra > /dev/null 2>&1
  python - $OUT/prof_overlap$v/b_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "grb::k_mxv" in r["Name"] or "k_long_" in r["Name"]:
        print("   ", r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
find $OUT -name '*kernel_trace.csv' -size +1M -delete
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "scale24" 2>&1 | tail -5
