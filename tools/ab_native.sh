#!/bin/bash
# GPU box: native-arithmetic FP16C kernels of this build against another build of the library (LUW_LIB), fresh process per measurement, interleaved; the
# tolerance gates of tests/test_gpu_native_arith.py first.   usage: tools/ab_native.sh <out dir> <other lib> [blocks...]
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$1"; OTHER="$2"; shift 2; mkdir -p "$O"
BLOCKS="${@:-c3_fp16c c3_fp16c_coriolis tile512_urban_fp16c_coriolis c5_rank_4x2x1_fp16c_coriolis}"
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python3 -m pytest "$R/tests/test_gpu_native_arith.py" -x -q > "$O/pytest_native.txt" 2>&1 || { tail -40 "$O/pytest_native.txt"; exit 1; }
  tail -3 "$O/pytest_native.txt"
fi
for rep in 1 2; do for blk in $BLOCKS; do for alt in "LUW_LIB=$OTHER" "X=1"; do for arith in native exact; do
  [ "$arith" = exact ] && [ "$alt" != "X=1" ] && continue
  flag=--secondary-block; case $blk in *rank*) flag=--rank-shape-block;; esac
  out=$(env $alt timeout -k 10 300 python3 "$R/bench.py" $flag $blk --arith $arith --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$blk [$arith, ${alt/X=1/this build}] $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" 2>&1 | tail -1)" | sed "s#$R/##" | tee -a "$O/ab_native.txt"
done; done; done; done
