#!/usr/bin/env bash
# GPU box: what each ingredient of the urban tile costs the FP16C pair kernel at 512^3 -- buildings, Coriolis (uniform-force instantiation), nudging + sponge zones
# (general instantiation) -- exact and native arithmetic; one fresh bench.py process per line.   usage: tools/urban_modes_probe.sh [rounds]
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
for r in $(seq 1 "${1:-1}"); do
  for args in "" "--buildings" "--buildings --coriolis" "--buildings --urban" "--buildings --urban --coriolis"; do
    for ar in exact native; do
      out=$(python3 "$R/bench.py" --workload tile512 --dtype fp16c $args --arith $ar --no-secondary --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1)
      echo "round $r tile512 fp16c [$args] $ar: $(python3 -c "import json,sys; j=json.loads(sys.argv[1]); print(j['ms_per_step'], 'ms/step  frac', j['roofline']['frac'], ' kernel', j['roofline'].get('kernel'))" "$out")"
    done
  done
done
