#!/usr/bin/env bash
# Runs the real reference binary (oracle/_ref/FluidX3D*) on one case directory and collects
# its console log + VTK outputs.  usage: run_ref_case.sh <binary> <case_dir> <out_dir> [timeout_s]
# Test infrastructure only (see oracle/build_ref.sh).
set -uo pipefail
BIN="$(readlink -f "$1")"; CASE="$(readlink -f "$2")"; OUT="$3"; TMO="${4:-300}"
mkdir -p "$OUT"; OUT="$(readlink -f "$OUT")"
WORK="$(mktemp -d)"
cp -r "$CASE"/. "$WORK"/
DECK="$(ls "$WORK"/*.luw* | head -1)"
( cd "$(dirname "$BIN")" && timeout "$TMO" "$BIN" "$DECK" </dev/null 2>&1 | sed 's/\x1b\[[0-9;]*[A-Za-z]//g' | tr '\r' '\n' > "$OUT/console.log" )
echo "exit=${PIPESTATUS[0]}" >> "$OUT/console.log"
if [ "${5:-}" != "novtk" ] && [ -d "$WORK/RESULTS/vtk" ]; then cp "$WORK"/RESULTS/vtk/*.vtk "$WORK"/RESULTS/*.csv "$OUT"/ 2>/dev/null; fi
ls -la "$OUT" | tail -n +2
rm -rf "$WORK"
