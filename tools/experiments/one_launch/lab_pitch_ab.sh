#!/bin/bash
# GPU box (lab): what the x pitch of a haloed rank domain costs -- the rank of [4,2,1] (514 cells per row: pitch 576 elements) as ONE whole-box launch with
# builds whose pitch is padded by 1, 2, 3, 7 blocks of 64 elements (640, 704, 768, 1024), and the undivided 512^3 urban tile with the same pads.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; : > "$O"
run() { # label, env..., -- bench args
  local label="$1"; shift
  env "$@" 2>/dev/null | python3 -c "
import json, sys
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-44s %.4f ms/step  kernel %s  frac %.4f' % ('$label', b['ms_per_step'], b.get('kernel_ms') or b['roofline'].get('kernel_ms'), b['roofline']['frac']))
" | tee -a "$O"
}
for rep in 1 2; do
  for pad in 0 1 2 3 7; do
    lib=(); [ $pad != 0 ] && lib=(LUW_LIB=$R/tools/lab_pitch$pad.so)
    run "c4 rank whole-box pad $pad" LUW_X_OVERLAP=0 "${lib[@]}" python3 $R/bench.py --rank-shape-block c4_rank_4x2x1_f32 --rank-transport peer-loopback --steps 100 --warmup 20
    run "c5 rank whole-box pad $pad" LUW_X_OVERLAP=0 "${lib[@]}" python3 $R/bench.py --rank-shape-block c5_rank_4x2x1_fp16c_coriolis --rank-transport peer-loopback --steps 100 --warmup 20
  done
  for pad in 0 1 2; do
    lib=(); [ $pad != 0 ] && lib=(LUW_LIB=$R/tools/lab_pitch$pad.so)
    run "tile512 urban f32 undivided pad $pad" "${lib[@]}" python3 $R/bench.py --secondary-block tile512_urban_f32 --steps 100 --warmup 20
  done
  run "c4 rank whole-box, x faces by pack kernels" LUW_X_OVERLAP=0 LUW_X_FACE_FUSED=0 python3 $R/bench.py --rank-shape-block c4_rank_4x2x1_f32 --rank-transport peer-loopback --steps 100 --warmup 20
done
