#!/bin/bash
# build host: scripts/build_variant.sh <name> -DFOO=1 -DBAR=2 ...   ->  build/variants/<name>/libgrb_mi355x.so (grb_mxv.hip with the defines, the other
# objects of the shipped library); run with GRB_MI355X_LIB=build/variants/<name>/libgrb_mi355x.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/python-graphblas_amd/csrc"
name=$1; shift
d="$ROOT/build/variants/$name"; mkdir -p "$d"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -Wno-unused-result -munsafe-fp-atomics "$@" -c "$SRC/grb_mxv.hip" -o "$d/grb_mxv.o" 2> "$d/build.log"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libgrb_mi355x.so" "$d/grb_mxv.o" $(ls "$SRC"/*.o | grep -v grb_mxv.o)
echo "built $d"
