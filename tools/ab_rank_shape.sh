#!/usr/bin/env bash
# Runs on the GPU box: one rank-shape block of bench.py (fresh process each time) under alternatives, interleaved in time.
# usage: tools/ab_rank_shape.sh <block> <rounds> <alt> ...   alt = "lib:<path to a libluw_core build>" or "env:NAME=VALUE[,NAME=VALUE]" or "base"
# A build under test is named to the process through LUW_LIB (capi.load picks it up); the product library in the package directory is never touched.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
BLOCK="$1"; ROUNDS="$2"; shift 2
for r in $(seq 1 "$ROUNDS"); do
  for alt in "$@"; do
    envs=()
    case "$alt" in lib:*) envs=("LUW_LIB=$R/${alt#lib:}");; env:*) IFS=, read -ra envs <<< "${alt#env:}";; esac
    out=$(env ${envs[@]+"${envs[@]}"} python3 "$R/bench.py" --rank-shape-block "$BLOCK" --steps 200 --warmup 20 2>/dev/null | tail -1)
    echo "$BLOCK round $r $alt: $(python3 -c "import json,sys; j=json.loads(sys.argv[1]); print(j['ms_per_step'], 'ms/step, interior kernel', j['kernel_ms'], 'shell', j.get('shell_ms'), 'exchange', j.get('exchange_ms'))" "$out")"
  done
done
