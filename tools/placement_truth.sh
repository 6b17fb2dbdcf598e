#!/usr/bin/env bash
# GPU box: does the probe of the placement search predict the step?  Per kind of allocation of the DDF array (search off): the probe the search WOULD measure
# (LUW_TUNE_PLACEMENT=1 LUW_TUNE_VERBOSE=1 prints it for the first candidate) and the kernel time of the real workload, fresh process each; then the search as
# shipped.   usage: tools/placement_truth.sh "<bench args>" [reps]     -> gpurun_out/placement_truth.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/placement_truth.txt"; mkdir -p "$R/gpurun_out"; : > "$OUT"
ARGS="$1"; REPS="${2:-2}"
one() {  # env assignments..., prints kernel ms, frac, placement
  env "$@" python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 40 --warmup 8 $ARGS 2> "$R/gpurun_out/placement_truth.err" | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['config']['placement']
print('kernel %.3f ms  frac %.4f  kept %s after %d candidates, probe %.2f TB/s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], p['kept'], p['candidates_tried'], p['probe_TBps']))"
  grep "placement candidate" "$R/gpurun_out/placement_truth.err" | sed 's/^/      /'
}
for r in $(seq "$REPS"); do
  for al in vmm:1024 vmm:2048 malloc vmm:512 vmm:4096 vmm:256; do
    echo "round $r  LUW_ALLOC=$al, no search: $(one LUW_ALLOC=$al LUW_TUNE_PLACEMENT=1 LUW_TUNE_VERBOSE=1)" >> "$OUT"
  done
  echo "round $r  as shipped: $(one LUW_TUNE_VERBOSE=1)" >> "$OUT"
done
cat "$OUT"
