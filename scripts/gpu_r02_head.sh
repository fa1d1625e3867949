#!/bin/bash
# short rows: persistent row groups with an LDS head (short_kernel 3) against the plain row groups (1), at thresholds
for cfg in "1 0" "3 0" "3 256" "3 128"; do
  set -- $cfg
  for wl in mxv_min_plus_masked mxv_lor_land_masked mxv_min_plus; do
    bash scripts/gpu_kt.sh r02u_sk$1_t$2_$wl $wl GRB_SHORT_KERNEL=$1 GRB_SPLIT_MIN_LEN=$2 2>&1 | grep -v "k_long_keys\|k_long_init\|k_pack\|k_x_image\|compact"
  done
done
