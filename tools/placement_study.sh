#!/usr/bin/env bash
# Runs on the GPU box: step-kernel time of fresh processes under different ways of allocating the DDF array
# (hipMalloc with / without the placement search, VMM-mapped chunks of several sizes).  Output: gpurun_out/placement_study.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/placement_study.txt"; : > "$OUT"
one() { # label, env assignments..., -- bench args
  local label="$1"; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local js; js=$(env "${envs[@]}" LUW_TUNE_VERBOSE=1 python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 60 --warmup 10 "$@" 2> /tmp/ps_err.txt | tail -1)
  local cand; cand=$(grep -c "placement candidate" /tmp/ps_err.txt || true)
  python3 - "$label" "$cand" "$js" >> "$OUT" <<'PY'
import json, sys
label, cand, js = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(js); print("%-44s kernel %.4f ms  frac %.4f  MLUPS %.0f  candidates %s" % (label, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["value"], cand))
except Exception as e:
    print("%-44s FAILED %s" % (label, str(e)[:100]))
PY
  tail -1 "$OUT"
}
for i in 1 2 3 4 5 6; do one "c2 malloc tune=0 #$i" LUW_TUNE_PLACEMENT=0 -- --workload c2; done
for c in 2 32 1024; do for i in 1 2 3; do one "c2 vmm:$c tune=0 #$i" LUW_TUNE_PLACEMENT=0 LUW_ALLOC=vmm:$c -- --workload c2; done; done
for i in 1 2; do one "c2 malloc tuned #$i" LUW_X=1 -- --workload c2; done
for i in 1 2 3 4; do one "c3 malloc tune=0 #$i" LUW_TUNE_PLACEMENT=0 -- --workload c3; done
for c in 2 32 1024; do for i in 1 2; do one "c3 vmm:$c tune=0 #$i" LUW_TUNE_PLACEMENT=0 LUW_ALLOC=vmm:$c -- --workload c3; done; done
for i in 1 2; do one "c3 malloc tuned #$i" LUW_X=1 -- --workload c3; done
for i in 1 2; do one "1024x1024x256 no buildings tuned #$i" LUW_X=1 -- --workload c2 --size 1024 1024 256; done
one "1024x1024x255 no buildings tuned" LUW_X=1 -- --workload c2 --size 1024 1024 255
one "c3 fp16c tuned" LUW_X=1 -- --workload c3 --dtype fp16c
one "c3 fp16c vmm:1024 tune=0" LUW_TUNE_PLACEMENT=0 LUW_ALLOC=vmm:1024 -- --workload c3 --dtype fp16c
