#!/bin/bash
# GPU box: rows per XCD and turn (LUW_XCD_ROWS=G; 0 = dispatch order) over a few lattice shapes.   usage: tools/sweep_xcd_rows_g.sh <out dir>
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$1"; mkdir -p "$O"; : > "$O/sweep_xcd_rows_g.txt"
for rep in 1 2; do for shp in 512,512,512 1024,1024,256 1024,1024,1024; do for dt in f32 fp16c; do for m in 0 1 2 4 16; do
  out=$(LUW_XCD_ROWS=$m timeout -k 10 300 python3 "$R/bench.py" --workload c2 --size ${shp//,/ } --dtype $dt --no-secondary --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1)
  echo "$shp $dt xcd_rows=$m $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tail -1)" | tee -a "$O/sweep_xcd_rows_g.txt"
done; done; done; done
