#!/usr/bin/env bash
# GPU box: size of the physical pieces lattice arrays are mapped from (LUW_ALLOC=vmm:<MiB>|vmm:one|malloc) against workload,
# REPS fresh processes each -> gpurun_out/chunk_study.txt.  usage: ALLOCS="vmm:1024 vmm:4096" tools/chunk_study.sh "<bench args>" ...
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/chunk_study.txt"; mkdir -p "$R/gpurun_out"; : > "$OUT"
python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --workload c2 2>/dev/null | tail -1 | python3 -c "import json,sys; print('device', json.loads(sys.stdin.read())['device'])" >> "$OUT"
for al in $ALLOCS; do
  for cfg in "$@"; do
    line="$al  $cfg :"
    for r in $(seq ${REPS:-1}); do
      ms=$(LUW_ALLOC=$al python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 40 --warmup 8 $cfg 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['roofline']; print('%.3f (%.3f)' % (d['kernel_ms'], d['frac']))")
      line="$line $ms"
    done
    echo "$line" >> "$OUT"
  done
done
cat "$OUT"
