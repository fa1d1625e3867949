#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compile the unmodified kernel sources for the CPU SIMT emulator -- with the flags of the shipped library
# (no -DGRB_ABLATE: the CPU tier exercises the same preprocessor path through every kernel as the GPU does).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
SRC="$ROOT/python-graphblas_amd/csrc"
OUT="$HERE/libgrb_emu.so"
mkdir -p "$HERE/obj"
pids=()
for f in "$SRC"/*.hip; do
  o="$HERE/obj/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find "$SRC" "$HERE/hip" "$HERE/rocprim" "$ROOT/include" -newer "$o" \( -name '*.hpp' -o -name '*.h' -o -name '*.inc' \) | head -1)" ]; then
    g++ -O1 -g -std=c++17 -fPIC -x c++ -I"$HERE" -I"$ROOT/include" -I"$SRC" -Wno-attributes -w -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
if [ ! -f "$HERE/obj/emu_runtime.o" ] || [ "$HERE/emu_runtime.cpp" -nt "$HERE/obj/emu_runtime.o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$HERE/obj/emu_runtime.o" ]; then
  g++ -O1 -g -std=c++17 -fPIC -I"$HERE" -c "$HERE/emu_runtime.cpp" -o "$HERE/obj/emu_runtime.o"
fi
# relink only when an object changed (several test processes may call this script at the same time)
if [ ! -f "$OUT" ] || [ -n "$(find "$HERE/obj" -name '*.o' -newer "$OUT" | head -1)" ]; then
  g++ -shared -o "$OUT.tmp.$$" "$HERE"/obj/*.o && mv -f "$OUT.tmp.$$" "$OUT"
fi
echo "built $OUT"
