#!/bin/bash
# GPU box: the FP16C rank of [4,2,1] with its x faces (a) written and read by the pair kernel (default), (b) written by the kernel, put by the insert kernel
# (LUW_X_INSERT_FUSED=0), (c) through pack and insert kernels (LUW_X_FACE_FUSED=0) -- default schedule and whole box + exchange, both transports.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; : > "$O"
for rep in 1 2; do for tr in peer-loopback rccl-self; do for ov in 1 0; do for x in "fused in+out" "LUW_X_INSERT_FUSED=0" "LUW_X_FACE_FUSED=0"; do
  e=(LUW_X_OVERLAP=$ov); [ "$x" != "fused in+out" ] && e+=("$x")
  env "${e[@]}" python3 $R/bench.py --rank-shape-block c5_rank_4x2x1_fp16c_coriolis --rank-transport $tr --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json, sys
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-13s overlap $ov  %-22s rep $rep  %.4f ms/step  kernel %s  shell %s  exchange %s  frac %.4f' % ('$tr', '$x', b['ms_per_step'], b['kernel_ms'], b['shell_ms'], b['exchange_ms'], b['roofline']['frac']))
" | tee -a "$O"
done; done; done; done
