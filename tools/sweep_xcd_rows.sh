#!/bin/bash
# GPU box: where does the workgroup order "a row's blocks on one XCD" (LUW_XCD_ROWS=1) pay?  Lattice shapes x {0, 1} x FP32 / FP16C, plain channel, fresh
# process per measurement, interleaved, two rounds.   usage: tools/sweep_xcd_rows.sh <out dir> [shapes "X,Y,Z ..."]
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$1"; mkdir -p "$O"; : > "$O/sweep_xcd_rows.txt"
SHAPES="${2:-512,512,512 1024,512,512 1024,1024,256 1024,1024,512 1024,1024,1024 2048,1024,512 2048,2048,256 512,512,2048}"
for rep in 1 2; do for shp in $SHAPES; do for dt in f32 fp16c; do for m in 0 1; do
  out=$(LUW_XCD_ROWS=$m timeout -k 10 300 python3 "$R/bench.py" --workload c2 --size ${shp//,/ } --dtype $dt --no-secondary --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1)
  echo "$shp $dt xcd_rows=$m $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f  %s' % (d['ms_per_step'], d['roofline']['frac'], d['config'].get('placement','')))" 2>&1 | tail -1)" | tee -a "$O/sweep_xcd_rows.txt"
done; done; done; done
