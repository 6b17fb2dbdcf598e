#!/usr/bin/env bash
# GPU box: ROUNDS times over all "ENV=VAL,ENV=VAL" settings (arguments) for each bench workload of WORKLOADS, interleaved in time so that a drift of
# the box does not pass for an effect of the setting -> gpurun_out/interleaved_ab.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/interleaved_ab.txt"; mkdir -p "$R/gpurun_out"; : > "$OUT"
IFS=';' read -ra WL <<< "${WORKLOADS:---workload c2;--workload c3}"
for round in $(seq ${ROUNDS:-3}); do
  for setting in "$@"; do
    for wl in "${WL[@]}"; do
      ms=$(env $(echo "$setting" | tr ',' ' ') python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 40 --warmup 8 $wl 2>/dev/null | tail -1 | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
      echo "round $round  $(date +%H:%M:%S)  $setting  [$wl]  $ms" >> "$OUT"
    done
  done
done
cat "$OUT"
