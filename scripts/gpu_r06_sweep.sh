#!/bin/bash
# Round 6: layout options of the headline re-swept on the round-6 library (the balance between the kernels moved since round 4)
OUT=gpurun_out/r06f; mkdir -p $OUT
one() { tag=$1; shift; env "$@" python bench.py --steps 30 --no-cpu-baseline --no-extra 2>$OUT/$tag.err | tee $OUT/$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), 'verified', d['verified'], 'build_ms', round(d['layout_build_call_ms'],1), 'cache_GB', round(d['preprocess_bytes']/1e9,2))"; }
one default A=1
one hub128 GRB_HUB_MIN_LEN=128
one hub256 GRB_HUB_MIN_LEN=256
one hub512 GRB_HUB_MIN_LEN=512
one hub2048 GRB_HUB_MIN_LEN=2048
one cls32 GRB_LONG_CLASSES=32
one cls32_hub256 GRB_LONG_CLASSES=32 GRB_HUB_MIN_LEN=256
one split32 GRB_SPLIT_MIN_LEN=32
one split32_hub256 GRB_SPLIT_MIN_LEN=32 GRB_HUB_MIN_LEN=256
one split128 GRB_SPLIT_MIN_LEN=128
one rt16k GRB_RTILE_ROWS=16384
one rt_e64k GRB_RTILE_ENTRIES=65536
one rt_e16k GRB_RTILE_ENTRIES=16384
one default2 A=1
