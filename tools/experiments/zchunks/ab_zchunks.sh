#!/bin/bash
# GPU box: the step pipelined along z (LUW_STEP_SCHEDULE=zchunks) against the default shell / interior schedule on the rank-shape blocks of bench.py (one rank
# of the 8-GPU tile in its real shape, whole step through RCCL's self send / receive or peer loopback), fresh process per measurement, interleaved.
#   usage: tools/ab_zchunks.sh <out dir> [blocks...]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; shift; mkdir -p "$O"; : > "$O/ab_zchunks.txt"
BLOCKS="${@:-c4_rank_4x2x1_f32 c5_rank_4x2x1_fp16c_coriolis}"
for rep in 1 2 3; do for blk in $BLOCKS; do for tr in rccl-self peer-loopback; do for sch in default zchunks; do
  if [ $sch = zchunks ]; then export LUW_STEP_SCHEDULE=zchunks; else unset LUW_STEP_SCHEDULE; fi
  out=$(timeout -k 10 300 python3 "$R/bench.py" --rank-shape-block $blk --rank-transport $tr --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$blk [$tr, $sch] $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f  kernel %s exchange %s' % (d['ms_per_step'], d['roofline']['frac'], d.get('kernel_ms'), d.get('exchange_ms')))" 2>&1 | tail -1)" | tee -a "$O/ab_zchunks.txt"
done; done; done; done
