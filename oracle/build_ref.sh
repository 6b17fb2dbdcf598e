#!/usr/bin/env bash
# Builds the REAL reference solver (LUW-modified FluidX3D) from the sources where they lie
# under /root/reference into oracle/_ref/FluidX3D. Test infrastructure only:
#   * nothing of the reference is copied into the repo (outputs: objects + one binary in oracle/_ref/,
#     which is git-ignored but travels to the GPU box with gpurun);
#   * the reference's own build system (make.sh / makefile) is NOT used: this is the same 11-file
#     object list (reference makefile:23) compiled with g++ directly;
#   * the binary links the system OpenCL ICD loader (libOpenCL.so.1), so on a box with a GPU it runs
#     the reference's OpenCL kernels on the AMD OpenCL runtime (libamdocl64) and can produce genuine
#     reference fields; in this CPU-only container it runs the host stage only (deck -> grid sizing).
# Skips silently (exit 0) when /root/reference is absent (GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${LUW_REFERENCE_ROOT:-/root/reference}/core/cfd_core/FluidX3D/src"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then echo "build_ref: $REF absent, keeping prebuilt oracle/_ref"; exit 0; fi
mkdir -p "$OUT/obj"
SRCS="graphics info kernel lbm lodepng main setup shapes fluxcorrection interpolation interpolation_hd"
CXXFLAGS="-std=c++17 -pthread -O -Wno-comment -w -I$REF/OpenCL/include"
pids=()
for s in $SRCS; do
  if [ ! -f "$OUT/obj/$s.o" ] || [ "$REF/$s.cpp" -nt "$OUT/obj/$s.o" ]; then
    g++ -c "$REF/$s.cpp" -o "$OUT/obj/$s.o" $CXXFLAGS &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
g++ $(for s in $SRCS; do echo "$OUT/obj/$s.o"; done) -o "$OUT/FluidX3D" -std=c++17 -pthread -O -lstdc++fs -lOpenCL
echo "build_ref: built $OUT/FluidX3D"
