#!/bin/bash
# GPU box: workgroup order experiment -- this build (round-robin: consecutive workgroups of a row on different XCDs) against the same sources with
# -DLUW_XCD_ROWS (the blocks of one row on ONE XCD, eight consecutive rows in flight), fresh process per measurement, interleaved.
#   usage: tools/ab_xcd_rows.sh <out dir> <lib built with -DLUW_XCD_ROWS> [blocks...]
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$1"; OTHER="$2"; shift 2; mkdir -p "$O"; : > "$O/ab_xcd_rows.txt"
BLOCKS="${@:-headline c2_f32 cube1024_f32 c3_fp16c cube1024_fp16c tile512_urban_fp16c_coriolis}"
for rep in 1 2 3; do for blk in $BLOCKS; do for alt in "X=1" "LUW_LIB=$OTHER"; do
  if [ $blk = headline ]; then args="--no-secondary --no-cpu-baseline"; else args="--secondary-block $blk"; fi
  out=$(env $alt timeout -k 10 300 python3 "$R/bench.py" $args --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$blk [${alt/X=1/round robin}] $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tail -1)" | sed "s#LUW_LIB=.*\]#rows per XCD]#" | tee -a "$O/ab_xcd_rows.txt"
done; done; done
