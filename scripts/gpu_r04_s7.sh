#!/bin/bash
cd "$(dirname "$0")/.."
echo "### default ranges"; WLS="mxv_min_plus_masked mxv_lor_land_masked" ORDERS="1 0" bash scripts/gpu_r04_variants.sh
for r in 8 16 48; do echo "### GRB_ORD_RANGES=$r"; GRB_ORD_RANGES=$r WLS="mxv_min_plus_masked" bash scripts/gpu_r04_variants.sh; done
echo "### GRB_ORD_RANGES=16 GRB_ORD_RANGE_KB=4096"; GRB_ORD_RANGES=16 GRB_ORD_RANGE_KB=4096 WLS="mxv_min_plus_masked" bash scripts/gpu_r04_variants.sh
echo "### GRB_ORD_RANGES=8 GRB_ORD_RANGE_KB=8192"; GRB_ORD_RANGES=8 GRB_ORD_RANGE_KB=8192 WLS="mxv_min_plus_masked" bash scripts/gpu_r04_variants.sh
