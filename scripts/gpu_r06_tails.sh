#!/bin/bash
# Round 6: the cold entries of the long rows below the hub level with the short rows (option cold_in_rows) -- parity on the GPU, then the headline,
# the unmasked call and the U[0,1) line with the option off / on and other hub limits; kernel times of off and on.
OUT=gpurun_out/r06i; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$VARIANT" ] && export GRB_MI355X_LIB=$PWD/build/variants/$VARIANT/libgrb_mi355x.so
python -m pytest tests/test_vertex_order.py -m gpu -x -q -k "sorted_row_tiles or ordered_product or shard_setup or ranked" 2>&1 | tail -3
one() { tag=$1; shift; env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'build_ms', round(d['layout_build_call_ms'],1), 'cache_GB', round(d['preprocess_bytes']/1e9,2), 'tails', d['stats'].get('long_tails'))"; }
oneu() { tag=$1; shift; env "$@" python bench.py --workload mxv_min_plus --steps 30 --no-cpu-baseline --no-extra 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'])"; }
one cold0 GRB_COLD_IN_ROWS=0
one cold1024 GRB_COLD_IN_ROWS=1024
one cold256 GRB_COLD_IN_ROWS=256
one cold2048_hub2048 GRB_COLD_IN_ROWS=2048 GRB_HUB_MIN_LEN=2048
one cold4096_hub4096 GRB_COLD_IN_ROWS=4096 GRB_HUB_MIN_LEN=4096
one cold512_hub512 GRB_COLD_IN_ROWS=512 GRB_HUB_MIN_LEN=512
one cold1024_e65536 GRB_COLD_IN_ROWS=1024 GRB_RTILE_ENTRIES=65536
one cold1024_e32768 GRB_COLD_IN_ROWS=1024 GRB_RTILE_ENTRIES=32768
one cold0_b GRB_COLD_IN_ROWS=0
oneu unmasked_cold0 GRB_COLD_IN_ROWS=0
oneu unmasked_cold1024 GRB_COLD_IN_ROWS=1024
for c in 0 1024; do
  GRB_COLD_IN_ROWS=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cold$c -o b -- python bench.py --steps 10 --no-cpu-baseline --no-extra > /dev/null 2>&1
  echo "kernels, cold_in_rows=$c"
  python - $OUT/prof_cold$c/b_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "grb::k_mxv" in r["Name"] or "k_long_init" in r["Name"] or "k_rows_tail" in r["Name"]:
        print("   ", r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
find $OUT -name '*kernel_trace.csv' -size +1M -delete
