#!/bin/bash
# A/B of build-time variants of grb_mxm.hip on the SpGEMM bench lines.
#   build host:  scripts/variants_mxm.sh build "GRB_MU_ILP=8" "GRB_MU_ILP_SYM=8" ...   (a variant may hold several defines: "A=1 -DB=2")
#   GPU box:     gpurun -- 'bash scripts/variants_mxm.sh run'
set -e; set +e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/python-graphblas_amd/csrc"
case "$1" in
build)
  shift
  make -C "$SRC" -j8 > /dev/null
  pids=()
  for def in "$@"; do
    d="$ROOT/build/variants/$(echo "$def" | tr ' ' '_')"; mkdir -p "$d"
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -Wno-unused-result -munsafe-fp-atomics -D$def \
        -c "$SRC/grb_mxm.hip" -o "$d/grb_mxm.o" 2> "$d/build.log" &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libgrb_mi355x.so" "$d/grb_mxm.o" $(ls "$SRC"/*.o | grep -v grb_mxm.o) &&
      echo "built $d" ) &
    pids+=($!)
    if [ ${#pids[@]} -ge 3 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  done
  wait ;;
run)
  cd "$ROOT"
  export TMPDIR=/tmp
  O=gpurun_out/variants_mxm; mkdir -p $O
  one() {
    timeout 600 python bench.py --workload mxm_plus_times --scale ${VARIANT_SCALE:-20} --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'verified', d['verified'])"
  }
  unset GRB_MI355X_LIB; one default
  for lib in build/variants/*/libgrb_mi355x.so; do
    [ -f "$lib" ] || continue
    tag="$(basename "$(dirname "$lib")")"
    GRB_MI355X_LIB="$ROOT/$lib" one "$tag"
    if [ -n "$VARIANT_PROFILE" ]; then
      (cd /tmp && GRB_MI355X_LIB="$ROOT/$lib" rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_$tag -o p -- python $ROOT/bench.py --workload mxm_plus_times --scale ${VARIANT_SCALE:-20} --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1)
      f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); grep "grb::k_spgemm" "$f" | head -6 | awk -F'","' '{printf "    %-70s calls %s avg_us %.0f\n", substr($1,2,70), $2, $4/1000}'
    fi
  done
  unset GRB_MI355X_LIB; one default ;;
*) echo "usage: $0 build <DEFINE=VALUE>... | run"; exit 2 ;;
esac
