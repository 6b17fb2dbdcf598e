#!/usr/bin/env bash
# Runs on the GPU box: one rank-shape block of bench.py (fresh process each time) under alternatives, interleaved in time.
# usage: tools/ab_rank_shape.sh <block> <rounds> <alt> ...   alt = "lib:<path to a libluw_core build>" or "env:NAME=VALUE[,NAME=VALUE]" or "base"
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
BLOCK="$1"; ROUNDS="$2"; shift 2
LIB="$R/latticeurbanwind_amd/csrc/libluw_core.so"; cp "$LIB" /tmp/luw_product.so; trap 'cp /tmp/luw_product.so "$LIB"' EXIT   # the product library is swapped in place per alternative and always put back
for r in $(seq 1 "$ROUNDS"); do
  for alt in "$@"; do
    cp /tmp/luw_product.so "$LIB"; envs=()
    case "$alt" in lib:*) cp "$R/${alt#lib:}" "$LIB";; env:*) IFS=, read -ra envs <<< "${alt#env:}";; esac
    out=$(env "${envs[@]}" python3 "$R/bench.py" --rank-shape-block "$BLOCK" --steps 200 --warmup 20 2>/dev/null | tail -1)
    echo "$BLOCK round $r $alt: $(python3 -c "import json,sys; j=json.loads(sys.argv[1]); print(j['ms_per_step'], 'ms/step, interior kernel', j['kernel_ms'], 'shell', j.get('shell_ms'), 'exchange', j.get('exchange_ms'))" "$out")"
  done
done
cp /tmp/luw_product.so "$LIB"
