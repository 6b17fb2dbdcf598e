#!/usr/bin/env bash
# Builds the reference solver in its documented FP32-DDF configuration ("//#define FP16C",
# FX/defines.hpp:13-14 -- the reference selects FP32 vs FP16 DDFs by editing that one line) into
# oracle/_ref/FluidX3D_fp32.  The sources are staged in a throw-away directory OUTSIDE the repo
# (mktemp -d), the one #define is commented out with sed, and the staging directory is deleted; nothing of the
# reference enters the repo or travels to the GPU box except the built binary.  Test infrastructure only.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${LUW_REFERENCE_ROOT:-/root/reference}/core/cfd_core/FluidX3D/src"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then echo "build_ref_fp32: $REF absent, keeping prebuilt oracle/_ref"; exit 0; fi
mkdir -p "$OUT"
STAGE="$(mktemp -d)"; trap 'rm -rf "$STAGE"' EXIT
cp "$REF"/*.cpp "$REF"/*.hpp "$STAGE"/
sed -i 's|^#define FP16C |//#define FP16C |' "$STAGE/defines.hpp"
grep -q '^//#define FP16C ' "$STAGE/defines.hpp"
SRCS="graphics info kernel lbm lodepng main setup shapes fluxcorrection interpolation interpolation_hd"
for s in $SRCS; do g++ -c "$STAGE/$s.cpp" -o "$STAGE/$s.o" -std=c++17 -pthread -O -w -I"$REF/OpenCL/include" & done
wait
g++ $(for s in $SRCS; do echo "$STAGE/$s.o"; done) -o "$OUT/FluidX3D_fp32" -std=c++17 -pthread -O -lstdc++fs -lOpenCL
echo "build_ref_fp32: built $OUT/FluidX3D_fp32"
