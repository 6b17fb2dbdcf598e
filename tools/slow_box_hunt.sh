#!/usr/bin/env bash
# GPU box: is this one of the boxes on which the 1024x1024x256 FP32 lattice misses the bar with every kind the placement search tries?  If so, study it while
# we are here: every kind of allocation (and plane skews) with the search off, real workload, fresh process each.   -> gpurun_out/slow_box_<host tag>.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/slow_box_$(date +%H%M%S).txt"; mkdir -p "$R/gpurun_out"
one() {
  env "$@" python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 30 --warmup 6 --workload c3 2> "$R/gpurun_out/slow_box.err" | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['config']['placement']
print('kernel %.3f ms  frac %.4f  kept %s after %d candidates, probe %.2f TB/s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], p['kept'], p['candidates_tried'], p['probe_TBps']))"
  grep "placement candidate" "$R/gpurun_out/slow_box.err" | sed 's/^/      /'
}
first=$(one LUW_TUNE_VERBOSE=1)
echo "as shipped: $first" | tee "$OUT"
ms=$(echo "$first" | sed -n 's/^kernel \([0-9.]*\) ms.*/\1/p')
if python3 -c "import sys; sys.exit(0 if float('$ms') > 6.95 else 1)"; then
  echo "SLOW BOX: studying" | tee -a "$OUT"
  "$R/tools/box_state_watch.sh" 14 | cut -c1-300 | sed 's/=\{5,\}[^;]*;//g' | tee -a "$OUT"        # the same kind again and again, with temperatures and clocks
  for al in vmm:1024 vmm:4096; do
    echo "LUW_ALLOC=$al, no search: $(one LUW_ALLOC=$al LUW_TUNE_PLACEMENT=0)" | tee -a "$OUT"
  done
  for sk in; do
    echo "LUW_PLANE_SKEW=$sk (1 GiB chunks), no search: $(one LUW_PLANE_SKEW=$sk LUW_TUNE_PLACEMENT=0)" | tee -a "$OUT"
    echo "LUW_PLANE_SKEW=$sk (2 GiB chunks), no search: $(one LUW_PLANE_SKEW=$sk LUW_ALLOC=vmm:2048 LUW_TUNE_PLACEMENT=0)" | tee -a "$OUT"
  done
  echo "as shipped again: $(one LUW_TUNE_VERBOSE=1)" | tee -a "$OUT"
else
  echo "ordinary box" | tee -a "$OUT"
fi
