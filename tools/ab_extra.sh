#!/usr/bin/env bash
# GPU box: rebuild the product library once per argument with EXTRA="<argument>" (compiler flags, e.g. "-DLUW_PAIR_WAVES=3"; "" = the
# product build), run the kernel parity tests (PARITY=0 skips them) and the benchmark lines of LINES (default: the FP16C ones).
# The product build is restored at the end.  Output: gpurun_out/ab_extra.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/ab_extra.txt"; mkdir -p "$R/gpurun_out"; : > "$OUT"
LINES="${LINES:-c2:fp16c c3:fp16c c3:fp16c:--coriolis cube1024:fp16c c2:f32 c3:f32}"
for extra in "$@"; do
  touch "$R/latticeurbanwind_amd/csrc/luw_core.hip"
  make -C "$R/latticeurbanwind_amd/csrc" -s EXTRA="$extra" >> "$OUT" 2>&1
  echo "== EXTRA='$extra'" >> "$OUT"
  if [ "${PARITY:-1}" = "1" ]; then
    python3 -m pytest "$R/tests/test_gpu_parity.py" "$R/tests/test_gpu_halo.py" -x -q 2>&1 | tail -3 >> "$OUT"
  fi
  for l in $LINES; do
    IFS=: read -r wl dt more <<< "$l"
    args="--workload $wl --dtype $dt ${more:-}"
    js=$(python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 60 --warmup 10 $args 2>/dev/null | tail -1)
    python3 - "$args" "$js" >> "$OUT" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print("%-50s kernel %.4f ms  frac %.4f  MLUPS %.0f" % (sys.argv[1], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"]))
except Exception as e:
    print("%-50s FAILED %s" % (sys.argv[1], str(e)[:80]))
PY
  done
done
touch "$R/latticeurbanwind_amd/csrc/luw_core.hip"; make -C "$R/latticeurbanwind_amd/csrc" -s >> "$OUT" 2>&1
cat "$OUT"
