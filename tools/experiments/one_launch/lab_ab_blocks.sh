#!/bin/bash
# GPU box (lab): bench blocks under alternative library builds (LUW_LIB), fresh process per run, interleaved.  usage: lab_ab_blocks.sh <out> <reps> "<blocks>" <lib|-> ...
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; REPS="$2"; BLOCKS="$3"; shift 3; : > "$O"
for rep in $(seq 1 "$REPS"); do for blk in $BLOCKS; do for lib in "$@"; do
  e=(); [ "$lib" != "-" ] && e=(LUW_LIB=$R/tools/$lib)
  flag=--secondary-block; case $blk in *rank*) flag=--rank-shape-block;; esac
  env "${e[@]}" python3 $R/bench.py $flag $blk --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json, sys
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-32s %-12s rep $rep  %.4f ms/step  kernel %s  frac %.4f' % ('$blk', '$lib', b['ms_per_step'], b.get('kernel_ms') or b['roofline'].get('kernel_ms'), b['roofline']['frac']))
" | tee -a "$O"
done; done; done
