#!/bin/bash
OUT=gpurun_out/r06g; mkdir -p $OUT
one() { tag=$1; shift; env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'build_ms', round(d['layout_build_call_ms'],1), 'cache_GB', round(d['preprocess_bytes']/1e9,2))"; }
one default A=1
for e in 40960 49152 65536 81920 98304 131072 196608; do one rt_e$e GRB_RTILE_ENTRIES=$e; done
for e in 65536 98304 131072; do one rt16k_e$e GRB_RTILE_ROWS=16384 GRB_RTILE_ENTRIES=$e; done
one default2 A=1

oneu() { tag=$1; shift; env "$@" python bench.py --workload mxv_min_plus --steps 30 --no-cpu-baseline --no-extra 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'])"; }
oneu unmasked_default A=1
oneu unmasked_e65536 GRB_RTILE_ENTRIES=65536
oneu unmasked_e131072 GRB_RTILE_ENTRIES=131072
oneb() { tag=$1; shift; env "$@" python bench.py --workload mxv_lor_land_masked --steps 30 --no-cpu-baseline --no-extra 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'])"; }
oneb bfs_default A=1
oneb bfs_e65536 GRB_RTILE_ENTRIES=65536
oneb bfs_e131072 GRB_RTILE_ENTRIES=131072
