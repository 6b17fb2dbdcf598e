#!/usr/bin/env bash
# GPU box: rebuild the product library with the pair kernel at 3 and 4 waves per SIMD, run the pair-kernel parity tests and the
# FP16C benchmark lines for each.  Output: gpurun_out/ab_pair_waves.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/ab_pair_waves.txt"; : > "$OUT"
for w in "$@"; do
  touch "$R/latticeurbanwind_amd/csrc/luw_core.hip"
  make -C "$R/latticeurbanwind_amd/csrc" -s EXTRA="-DLUW_PAIR_WAVES=$w" >> "$OUT" 2>&1
  echo "== waves $w" >> "$OUT"
  if [ "${PARITY:-1}" = "1" ]; then
    python3 -m pytest "$R/tests/test_gpu_parity.py" "$R/tests/test_gpu_halo.py" -x -q -k "fp16c or True or pair or p-" 2>&1 | tail -3 >> "$OUT"
  fi
  for args in "--workload c2 --dtype fp16c" "--workload c3 --dtype fp16c" "--workload c3 --dtype fp16c --coriolis" "--workload cube1024 --dtype fp16c --steps 30"; do
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
