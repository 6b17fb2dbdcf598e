#!/usr/bin/env bash
# GPU box: does the step rate follow the device's state over time rather than the kind of allocation?  The headline lattice (1024x1024x256 FP32, buildings) on
# ONE kind (1 GiB chunks, search off), fresh process each, back to back for a few minutes, with the temperatures / clocks / power rocm-smi reports around
# each run.   usage: tools/box_state_watch.sh [runs]   -> gpurun_out/box_state_<time>.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$R/gpurun_out/box_state_$(date +%H%M%S).txt"; mkdir -p "$R/gpurun_out"
smi() { rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "Temperature|Power|sclk|mclk|fclk" | sed 's/^GPU\[[0-9]*\][ \t]*: //' | tr '\n' ';' | cut -c1-400; }
rocm-smi --showtemp --showpower --showclocks 2>&1 | head -40 > "$R/gpurun_out/box_state_smi_raw.txt"
for i in $(seq "${1:-12}"); do
  before=$(smi)
  ms=$(LUW_TUNE_PLACEMENT=0 python3 "$R/bench.py" --no-secondary --no-cpu-baseline --steps 60 --warmup 10 --workload c3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))")
  echo "run $i  $(date +%H:%M:%S)  $ms  | before: $before | after: $(smi)" | tee -a "$OUT"
done
