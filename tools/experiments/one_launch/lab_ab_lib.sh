#!/bin/bash
# GPU box (lab): rank-shape blocks under alternative library builds, whole-box schedule, peer-loopback, interleaved.  usage: lab_ab_lib.sh <out> <blocks> <lib|-> ...
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; BLOCKS="$2"; shift 2; : > "$O"
for rep in 1 2 3; do for blk in $BLOCKS; do for lib in "$@"; do
  e=(); [ "$lib" != "-" ] && e=(LUW_LIB=$R/tools/$lib)
  env LUW_X_OVERLAP=${LUW_X_OVERLAP_AB:-0} "${e[@]}" python3 $R/bench.py --rank-shape-block $blk --rank-transport peer-loopback --steps 100 --warmup 20 2>/dev/null | python3 -c "
import json, sys
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-30s %-18s rep $rep  %.4f ms/step  kernel %s  frac %.4f' % ('$blk', '$lib', b['ms_per_step'], b['kernel_ms'], b['roofline']['frac']))
" | tee -a "$O"
done; done; done
