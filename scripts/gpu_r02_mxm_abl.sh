#!/bin/bash
# round 2: where the numeric unit kernels of the SpGEMM spend their time (-DGRB_ABLATE build; results are wrong on purpose)
# bits (<< 20): 1 no column writes, 2 no value writes, 4 no accumulation, 8 no pass B at all, 16 no pass A in the numeric pass
# usage: gpu_r02_mxm_abl.sh SCALE FLAGS...   (per-kernel times of every run)
export GRB_MI355X_LIB=$PWD/build/ablate/libgrb_mi355x.so
SC=$1; shift
for f in "$@"; do
  bash scripts/gpu_kt_mxm.sh mxmabl_$f $SC GRB_DEBUG_FLAGS=$((f << 20)) 2>&1 | cut -c1-140 | head -8
done
