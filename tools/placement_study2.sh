#!/usr/bin/env bash
# Runs on the GPU box: consistency of the default allocation (arrays mapped from 1 GiB physical chunks) over fresh processes, and
# other chunk sizes.  Output: gpurun_out/placement_study2.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/placement_study2.txt"; : > "$OUT"
one() {
  local label="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local js; js=$(env "${envs[@]}" LUW_TUNE_VERBOSE=1 python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 60 --warmup 10 "$@" 2> /tmp/ps_err.txt | tail -1)
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
for i in 1 2 3 4 5 6 7 8 9 10; do one "c2 default (vmm:1024) #$i" LUW_X=1 -- --workload c2; done
for i in 1 2 3 4 5; do one "c3 default (vmm:1024) #$i" LUW_X=1 -- --workload c3; done
for c in 256 4096 131072; do for i in 1 2; do one "c2 vmm:$c #$i" LUW_ALLOC=vmm:$c -- --workload c2; one "c3 vmm:$c #$i" LUW_ALLOC=vmm:$c -- --workload c3; done; done
one "c2 fp16c default" LUW_X=1 -- --workload c2 --dtype fp16c
one "c3 fp16c default" LUW_X=1 -- --workload c3 --dtype fp16c
one "c3 fp16c vmm:131072" LUW_ALLOC=vmm:131072 -- --workload c3 --dtype fp16c
one "cube1024 f32 default" LUW_X=1 -- --workload cube1024 --steps 30
one "cube1024 fp16c default" LUW_X=1 -- --workload cube1024 --dtype fp16c --steps 30
one "cube1024 f32 vmm:131072" LUW_ALLOC=vmm:131072 -- --workload cube1024 --steps 30
one "c2 malloc tuned" LUW_ALLOC=malloc -- --workload c2
