#!/bin/bash
# Round 4, GPU call 5: build-time ablations of k_mxv_rows_tag / k_mxv_ctile on the ordered layouts (results wrong on purpose): what the
# gathers, the LDS atomics and the tile flush cost.
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
for lib in "" build/variants/*/libgrb_mi355x.so; do
  tag=default; [ -n "$lib" ] && tag=$(basename $(dirname $lib))
  for wl in mxv_min_plus_masked mxv_min_plus; do
    ( cd /tmp && GRB_MI355X_LIB=${lib:+$ROOT/$lib} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_${tag}_$wl" -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-extra --workload $wl --steps 10 > /tmp/b.json 2> /tmp/b.err )
    f=$(find /tmp/prof_${tag}_$wl -name '*kernel_stats.csv' | head -1)
    echo "== $tag $wl $(python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],4))")"
    grep -E "grb::k_mxv_(rows_tag|ctile|hstrip)" "$f" | awk -F'",' '{split($2,a,","); printf "   %-50s avg %.1f us\n", substr($1,12,50), a[3]/1000}'
  done
done
