#!/usr/bin/env bash
# GPU box: kernel time of the empty-channel workload over lattice shapes ("Nx Ny Nz" per argument; DTYPE=f32|fp16c) -> gpurun_out/size_study.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/size_study.txt"; mkdir -p "$R/gpurun_out"
for sz in "$@"; do
  python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 40 --warmup 8 --workload c2 --dtype ${DTYPE:-f32} --size $sz 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-16s %s ${LUW_ALLOC:-} kernel %.4f ms frac %.4f MLUPS %.0f' % ('$sz', '${DTYPE:-f32}', d['roofline']['kernel_ms'], d['roofline']['frac'], d['value']))" >> "$OUT"
done
cat "$OUT"
