#!/bin/bash
# GPU box (lab): ns per cell of the FP32 step on undivided empty channels whose x / y extents are those of a haloed rank domain, against 512^3
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; : > "$O"
for rep in 1 2; do for sz in "512 512 512" "512 514 512" "514 512 512" "514 514 512" "576 512 512" "512 576 512" "512 512 514"; do
  python3 $R/bench.py --workload c2 --size $sz --steps 100 --warmup 20 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); n = 1
for v in b['config']['global_lattice']: n *= v
print('%-14s rep $rep  kernel %.4f ms  %.5f ns/cell  frac %.4f' % ('$sz', b['roofline']['kernel_ms'], b['roofline']['kernel_ms'] * 1e6 / n, b['roofline']['frac']))
" | tee -a "$O"
done; done
python3 $R/tools/box_rate_probe.py f32 2>/dev/null | grep -i "whole\|y-half\|z-quarter" | tee -a "$O"
