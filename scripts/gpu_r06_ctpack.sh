#!/bin/bash
# Round 6: packed row-tile words (option ctile_pack): parity on the GPU, headline / unmasked with the option on and off, kernel times
OUT=gpurun_out/r06v; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$VARIANT" ] && export GRB_MI355X_LIB=$PWD/build/variants/$VARIANT/libgrb_mi355x.so
python -m pytest tests/test_vertex_order.py tests/test_gpu_scale.py -m gpu -x -q -k "sorted_row_tiles or ordered_product or scale24_headline or rmat_vs_oracle" 2>&1 | tail -3
one() { tag=$1; shift; env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra --no-warm-build 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'verified', d['verified'], 'cache_GB', round(d['preprocess_bytes']/1e9,3))"; }
oneu() { tag=$1; shift; env "$@" python bench.py --workload mxv_min_plus --steps 30 --no-cpu-baseline --no-extra --no-warm-build 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'])"; }
one cp1 GRB_CTILE_PACK=1; one cp0 GRB_CTILE_PACK=0; one cp1_b GRB_CTILE_PACK=1; one cp0_b GRB_CTILE_PACK=0
one cp2 GRB_CTILE_PACK=2; one cp2_b GRB_CTILE_PACK=2
oneu unmasked_cp1 GRB_CTILE_PACK=1; oneu unmasked_cp0 GRB_CTILE_PACK=0
for c in 1 2 0; do
  GRB_CTILE_PACK=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cp$c -o b -- python bench.py --steps 10 --no-cpu-baseline --no-extra --no-warm-build > /dev/null 2>&1
  echo "kernels, ctile_pack=$c"
  python - $OUT/prof_cp$c/b_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "grb::k_mxv_r" in r["Name"] or "hstrip" in r["Name"] or "ctile" in r["Name"]:
        print("   ", r["Name"][:76].ljust(76), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
find $OUT -name '*kernel_trace.csv' -size +1M -delete
