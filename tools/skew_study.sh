#!/usr/bin/env bash
# GPU box: plane skew (LUW_PLANE_SKEW, in 256-byte blocks) against lattice shape, REPS fresh processes each -> gpurun_out/skew_study.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/skew_study.txt"; mkdir -p "$R/gpurun_out"; : > "$OUT"
for sk in $SKEWS; do
  for cfg in "$@"; do
    line="skew $sk  $cfg :"
    for r in $(seq ${REPS:-3}); do
      ms=$(LUW_PLANE_SKEW=$sk python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | tail -1 | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
      line="$line $ms"
    done
    echo "$line" >> "$OUT"
  done
done
cat "$OUT"
