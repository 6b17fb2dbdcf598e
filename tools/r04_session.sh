#!/usr/bin/env bash
# GPU box, round 4: the measurements behind DESIGN.md "Status after round 4" in one session (interleaved where it matters).  Parts: repro rank place sweep
# usage: tools/r04_session.sh <part> ...       outputs under gpurun_out/r04_*.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
O="$R/gpurun_out"; mkdir -p "$O"
for part in "$@"; do case "$part" in
repro)   # how a multi-chunk mapping wants to be torn down (tools/vmm_unmap_repro.hip)
  hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_unmap_repro "$R/tools/vmm_unmap_repro.hip" 2>/dev/null
  { echo "== per chunk"; timeout -k 10 120 /tmp/vmm_unmap_repro chunk 4; echo "exit $?"; echo "== one hipMemUnmap over the range (round 3)"; timeout -k 10 120 /tmp/vmm_unmap_repro whole 4; echo "exit $?";
    echo "== the placement search's sequence (1 GiB chunks kept; 2 GiB chunks, hipMalloc, 512 MiB chunks tried and released), per chunk"; timeout -k 10 120 /tmp/vmm_unmap_repro search-chunk 3; echo "exit $?";
    echo "== the same, whole range"; timeout -k 10 120 /tmp/vmm_unmap_repro search-whole 3; echo "exit $?"; } > "$O/r04_vmm_unmap_repro.txt" 2>&1
  ;;
rank)    # one rank of the literal [4,2,1] cut: x faces from the step kernels against the pack kernel, slab thickness, transport
  { for blk in c4_rank_4x2x1_f32 c5_rank_4x2x1_fp16c_coriolis; do
      bash "$R/tools/ab_rank_shape.sh" $blk 2 base env:LUW_X_FACE_FUSED=0 env:LUW_X_SHELL=64 env:LUW_X_SHELL=32 env:BENCH_RANK_TRANSPORT=peer-loopback env:BENCH_RANK_TRANSPORT=peer-loopback,LUW_X_SHELL=64
    done; } > "$O/r04_rank_shape_ab.txt" 2>&1
  ;;
place)   # ten fresh processes per lattice: what the bounded placement search keeps, and how far the step times spread
  { for wl in c2 c3; do for i in 1 2 3 4 5 6 7 8 9 10; do
      python3 "$R/bench.py" --workload $wl --no-secondary --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['config']['placement']
print('$wl run $i: kernel %.4f ms  frac %.4f  create %.2f s  placement kept %s after %d candidates (probe %.2f TB/s)' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['create_s'], p['kept'], p['candidates_tried'], p['probe_TBps']))"
    done; done; } > "$O/r04_placement_10x.txt" 2>&1
  ;;
sweep)   # length of the timed region (the driver times 20 steps, the secondary blocks 200)
  rm -f "$O/steps_sweep.txt"; bash "$R/tools/steps_sweep.sh" "" 20 50 200 1000; mv "$O/steps_sweep.txt" "$O/r04_steps_sweep.txt"
  ;;
esac; done
