#!/bin/bash
# A/B of build-time variants of libgrb_mi355x.so on the bench workloads.
#   on the build host:   scripts/variants_ab.sh build "GRB_GATHER_AUX=2" "GRB_GATHER_AUX=1" "GRB_ROWS_EPL=8"
#   on the GPU box:      gpurun -- 'bash scripts/variants_ab.sh run'
# `build` compiles grb_mxv.hip once per define into build/variants/<define>/libgrb_mi355x.so (git-ignored, travels with gpurun);
# `run` benches the default library and every variant (masked min_plus + the --extra workloads), default first and last.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/python-graphblas_amd/csrc"
case "$1" in
build)
  shift
  make -C "$SRC" -j8 > /dev/null
  for def in "$@"; do
    d="$ROOT/build/variants/$def"; mkdir -p "$d"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -Wno-unused-result -munsafe-fp-atomics "-D$def" \
        -c "$SRC/grb_mxv.hip" -o "$d/grb_mxv.o"
    objs=$(ls "$SRC"/*.o | grep -v grb_mxv.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libgrb_mi355x.so" "$d/grb_mxv.o" $objs
    echo "built $d/libgrb_mi355x.so"
  done ;;
run)
  cd "$ROOT"
  one() {
    python bench.py --steps 30 --no-cpu-baseline --no-extra ${VARIANT_BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), 'verified', d['verified'])"
  }
  unset GRB_MI355X_LIB; one default
  for lib in build/variants/*/libgrb_mi355x.so; do
    [ -f "$lib" ] || continue
    GRB_MI355X_LIB="$ROOT/$lib" one "$(basename "$(dirname "$lib")")"
  done
  unset GRB_MI355X_LIB; one default ;;
*) echo "usage: $0 build <DEFINE=VALUE>... | run"; exit 2 ;;
esac
