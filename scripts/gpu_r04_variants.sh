#!/bin/bash
# Round 4: the shipped library and every build/variants/*/libgrb_mi355x.so on the headline (masked) and the unmasked relaxation, with the
# steady-state kernel times from rocprofv3.   WLS="..." ORDERS="1 0" bash scripts/gpu_r04_variants.sh
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
WLS=${WLS:-mxv_min_plus_masked mxv_min_plus}
for lib in "" build/variants/*/libgrb_mi355x.so; do
  tag=default; [ -n "$lib" ] && tag=$(basename $(dirname $lib))
  for om in ${ORDERS:-1}; do
    [ "$om" = "0" ] && [ -n "$lib" ] && [ -z "$ALL_ORDERS" ] && continue
    for wl in $WLS; do
      ( cd /tmp && GRB_ORDER_MODE=$om GRB_MI355X_LIB=${lib:+$ROOT/$lib} timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_${tag}_${om}_$wl" -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-extra --workload $wl --steps 10 > /tmp/b.json 2> /tmp/b.err )
      f=$(find /tmp/prof_${tag}_${om}_$wl -name '*kernel_stats.csv' | head -1)
      echo "== $tag order=$om $wl $(python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],4), d['verified'], round(d.get('layout_build_call_ms',0),1))" 2>&1 | tail -1)"
      grep -E "grb::k_mxv_(rows_tag|ctile|hstrip|stile|long_grp)" "$f" | awk -F'",' '{split($2,a,","); printf "   %-50s avg %.1f us\n", substr($1,12,50), a[3]/1000}'
    done
  done
done
