#!/bin/bash
# Round 4, GPU call 2: k_mxv_rows_tag with every cold gather folded into the first 2 MiB of the image (GRB_TAG_ABL=128; results wrong):
# what the short rows cost when no gather misses L2 -- the upper bound of moving their cold entries into column-range tiles.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_s2; mkdir -p "$OUT"; export TMPDIR=/tmp
ROOT=$PWD
run() {  # run <tag> <lib or ""> <bench args>
  tag=$1; lib=$2; shift 2
  ( cd /tmp && GRB_MI355X_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$tag" -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-extra "$@" > $ROOT/$OUT/bench_$tag.json 2> $ROOT/$OUT/prof_$tag.err )
  cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$tag.csv
  echo "== $tag"; python -c "import json;d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['verified'])"
  grep -E "grb::k_mxv|grb::k_x_image|grb::k_long" "$OUT/kernel_stats_$tag.csv" | grep -v "pull\|seams" | awk -F'",' '{split($2,a,","); printf "   %-70s %s us\n", substr($1,2,70), a[3]/1000}'
}
run default_masked ""
run fold_masked "$ROOT/build/variants/GRB_TAG_ABL=128/libgrb_mi355x.so"
run default_unmasked "" --workload mxv_min_plus
run fold_unmasked "$ROOT/build/variants/GRB_TAG_ABL=128/libgrb_mi355x.so" --workload mxv_min_plus
run default_bfs "" --workload mxv_lor_land_masked
run fold_bfs "$ROOT/build/variants/GRB_TAG_ABL=128/libgrb_mi355x.so" --workload mxv_lor_land_masked
