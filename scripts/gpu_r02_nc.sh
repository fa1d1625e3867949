#!/bin/bash
# strips: number of classes x long-row threshold x hot-table size; per-kernel times for the masked and unmasked min_plus call
for cfg in "8 64 0" "16 64 655360" "16 128 655360" "32 128 1310720" "16 32 655360"; do
  set -- $cfg
  for wl in mxv_min_plus_masked mxv_min_plus; do
    bash scripts/gpu_kt.sh r02m_nc$1_t$2_$wl $wl GRB_LONG_CLASSES=$1 GRB_SPLIT_MIN_LEN=$2 GRB_HOT_K=$3 2>&1 | grep -v "k_long_keys\|k_long_init\|k_pack"
  done
done
