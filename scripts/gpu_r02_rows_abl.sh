#!/bin/bash
# rows kernel: what the gather misses cost (-DGRB_ABLATE build; results WRONG): gathers folded into the first 2^19 / 2^15 image entries
export GRB_MI355X_LIB=$PWD/build/ablate/libgrb_mi355x.so
for f in 0 16384 32768; do
  bash scripts/gpu_kt.sh rows_abl_$f mxv_min_plus_masked GRB_DEBUG_FLAGS=$f 2>&1 | grep -E "^==|k_mxv_rows|k_mxv_strip"
done
