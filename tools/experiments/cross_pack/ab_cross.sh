#!/bin/bash
# GPU box: the FP16C pair kernel with both cells of a lane at once (LUW_PAIR_CROSS=1) against the one-after-the-other kernels: bit-for-bit tests first,
# then fresh-process bench blocks interleaved.   usage: tools/experiments/cross_pack/ab_cross.sh <out dir> [blocks...]   (after applying product_sources.patch)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; shift; mkdir -p "$O"
BLOCKS="${@:-c2_fp16c c3_fp16c c3_fp16c_coriolis tile512_urban_fp16c_coriolis}"
if [ -z "$SKIP_TESTS" ]; then
  LUW_PAIR_CROSS=1 timeout -k 10 900 python3 -m pytest "$R/tests/test_gpu_parity.py" "$R/tests/test_gpu_halo.py" -x -q -k "fp16 or pair or force or zone or halo or coriolis" > "$O/pytest_cross.txt" 2>&1 \
    || { tail -30 "$O/pytest_cross.txt"; exit 1; }
  tail -3 "$O/pytest_cross.txt"
fi
for rep in 1 2; do for blk in $BLOCKS; do for alt in "LUW_PAIR_CROSS=0" "LUW_PAIR_CROSS=1" "LUW_PAIR_CROSS=1 LUW_LIB=$R/tools/libluw_core_w3.so"; do
  flag=--secondary-block; case $blk in *rank*) flag=--rank-shape-block;; esac
  out=$(env $alt timeout -k 10 300 python3 "$R/bench.py" $flag $blk --steps 200 --warmup 20 2>/dev/null | tail -1)
  echo "$blk [$alt] $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  frac %.4f  kernel_ms %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms', d.get('kernel_ms'))))" 2>&1 | tail -1)" | sed "s#$R/##" | tee -a "$O/ab_cross.txt"
done; done; done
