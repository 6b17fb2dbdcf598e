#!/usr/bin/env bash
# Static instruction statistics of the stream_collide kernels: compiles luw_core.hip with --save-temps into /tmp/isa and
# prints, per kernel symbol matching $1 (default: the product kernels), VALU / SALU / memory instruction counts, 64-bit
# address adds, VGPRs and occupancy.  usage: tools/isa_stats.sh [symbol-regex] [extra hipcc flags...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; PAT="${1:-k_stream_collide_sI[tf]Li1ELi[04]ELi2ELb[01]E|k_stream_collide_pILi1ELi0E}"; shift || true
mkdir -p /tmp/isa && cd /tmp/isa
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -I"$ROOT/include" --save-temps -c -o luw.o "$ROOT/latticeurbanwind_amd/csrc/luw_core.hip" "$@" 2>/dev/null
S=luw_core-hip-amdgcn-amd-amdhsa-gfx950.s
for k in $(grep -o "^_Z[A-Za-z0-9_]*:" $S | tr -d : | grep -E "$PAT"); do
  awk -v k="^$k:" '$0~k{p=1} p{print} /^\.Lfunc_end/{if(p)exit}' $S > "k_$k.s"
  printf "%-60s valu %4d salu %4d mem %3d add64 %2d  " "$k" "$(grep -c '^\s*v_' k_$k.s)" "$(grep -c '^\s*s_' k_$k.s)" "$(grep -c '^\s*global_' k_$k.s)" "$(grep -c 'v_lshl_add_u64' k_$k.s)"
  awk -v k="$k" '$0~"^; Kernel info|^; codeLenInByte"{next} p&&/; NumVgprs:/{v=$3} p&&/; Occupancy:/{print "vgpr " v " occupancy " $3; exit} $0~"^"k":"{p=1}' $S
done
