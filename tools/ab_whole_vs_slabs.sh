#!/bin/bash
# GPU box: one rank of the literal cut [4,2,1] (BASELINE configs[3] / [4]) stepped (a) with the x slabs + shell / interior overlap and (b) as ONE whole-box
# launch followed by the exchange (LUW_X_OVERLAP=0), fresh process per run, interleaved, both transports.   usage: tools/ab_whole_vs_slabs.sh <out file> [reps]
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$1"; REPS="${2:-2}"; : > "$O"
for rep in $(seq 1 "$REPS"); do for blk in c4_rank_4x2x1_f32 c5_rank_4x2x1_fp16c_coriolis; do for tr in rccl-self peer-loopback; do for mode in slabs whole; do
  if [ $mode = slabs ]; then unset LUW_X_OVERLAP; else export LUW_X_OVERLAP=0; fi
  timeout -k 10 300 python3 "$R/bench.py" --rank-shape-block $blk --rank-transport $tr --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json, sys
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-30s %-13s %-6s rep $rep  %.4f ms/step  kernel %s  shell %s  exchange %s  frac %.4f  overlap %s' % ('$blk', '$tr', '$mode', b['ms_per_step'], b['kernel_ms'], b['shell_ms'], b['exchange_ms'], b['roofline']['frac'], b['overlap']))
" | tee -a "$O"
done; done; done; done
