#!/usr/bin/env bash
# GPU box: chunk size of the mapped lattice arrays against the plane size (512^3: 0.5 GiB planes, 1024x1024x256: 1 GiB, 1024^3: 4 GiB)
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/placement_study3.txt"; : > "$OUT"
one() {
  local label="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local js; js=$(env "${envs[@]}" python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 40 --warmup 8 "$@" 2> /tmp/ps_err.txt | tail -1)
  python3 - "$label" "$js" >> "$OUT" <<'PY'
import json, sys
label, js = sys.argv[1], sys.argv[2]
try:
    d = json.loads(js); print("%-44s kernel %.4f ms  frac %.4f  MLUPS %.0f" % (label, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"]))
except Exception as e:
    print("%-44s FAILED %s" % (label, str(e)[:100]))
PY
  tail -1 "$OUT"
}
for c in 2 32 256 512 2048; do one "cube1024 f32 vmm:$c" LUW_ALLOC=vmm:$c -- --workload cube1024 --steps 20; done
for c in 128 256 512; do one "c3 f32 vmm:$c" LUW_ALLOC=vmm:$c -- --workload c3; one "c2 f32 vmm:$c" LUW_ALLOC=vmm:$c -- --workload c2; done
one "cube1024 fp16c vmm:256" LUW_ALLOC=vmm:256 -- --workload cube1024 --dtype fp16c --steps 20
one "cube1024 fp16c vmm:1024" LUW_ALLOC=vmm:1024 -- --workload cube1024 --dtype fp16c --steps 20
