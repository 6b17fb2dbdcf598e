#!/usr/bin/env bash
# GPU box: mean stream_collide kernel time against the NUMBER OF TIMED STEPS of one bench.py run (fresh process per line): does a longer timed region
# run slower (clocks / power) than a short one?  usage: steps_sweep.sh "<bench args>" steps...   -> gpurun_out/steps_sweep.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/steps_sweep.txt"; mkdir -p "$R/gpurun_out"
ARGS="$1"; shift
for rep in 1 2; do
  for k in "$@"; do
    ms=$(python3 "$R/bench.py" --no-secondary --no-cpu-baseline --no-parity --steps $k --warmup 10 $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f kernel  %.4f wall' % (d['roofline']['kernel_ms'], d['ms_per_step']))")
    echo "rep $rep  [$ARGS]  steps $k  $ms" >> "$OUT"
  done
done
