#!/usr/bin/env bash
# GPU box: the default bench sequence (headline, then the secondary blocks in one process) under each "ENV=VAL,..." setting, ROUNDS times interleaved
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/bench_sequence_ab.txt"; mkdir -p "$R/gpurun_out"; : > "$OUT"
for round in $(seq ${ROUNDS:-2}); do
  for setting in "$@"; do
    env $(echo "$setting" | tr ',' ' ') python3 "$R/bench.py" --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('round $round $setting : c3_f32 %.3f' % d['roofline']['kernel_ms'], ' '.join('%s %.3f' % (k, v['roofline']['kernel_ms']) for k, v in d['secondary'].items() if 'roofline' in v))" >> "$OUT"
  done
done
cat "$OUT"
