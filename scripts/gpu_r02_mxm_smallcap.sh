#!/bin/bash
# SpGEMM: capacity of the one-wavefront class (build variants -DGRB_MU_SMALL=256 / 1024 under build/variants)
one() { python bench.py --workload mxm_plus_times --scale 20 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2))"; }
one default
GRB_MI355X_LIB=$PWD/build/variants/GRB_MU_SMALL=1024/libgrb_mi355x.so GRB_MXM_UNIT_SMALL=1024 GRB_MXM_UNIT_MID=1024 one "small=1024,no-mid"
GRB_MI355X_LIB=$PWD/build/variants/GRB_MU_SMALL=1024/libgrb_mi355x.so GRB_MXM_UNIT_SMALL=768 GRB_MXM_UNIT_MID=1024 one "cap1024,small=768"
GRB_MI355X_LIB=$PWD/build/variants/GRB_MU_SMALL=256/libgrb_mi355x.so one "small=256"
GRB_MI355X_LIB=$PWD/build/variants/GRB_MU_SMALL=256/libgrb_mi355x.so GRB_MXM_UNIT_MID=512 one "small=256,mid=512"
one default
