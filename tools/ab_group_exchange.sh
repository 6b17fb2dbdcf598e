#!/bin/bash
# GPU box: the one-process host's exchange routes against each other -- fuzz of both (and the oracle) first, then tools/bench_group.py's eight-domain cases
# on the one GPU under the one-phase default and LUW_GROUP_EXCHANGE=sequential, FP32 and FP16C.   usage: tools/ab_group_exchange.sh <out dir> [fuzz cases]
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$1"; mkdir -p "$O"
timeout -k 10 900 python3 "$R/tests/fuzz/fuzz_exchange_group_gpu.py" "${2:-500}" > "$O/fuzz_group.txt" 2>&1; rc=$?; tail -2 "$O/fuzz_group.txt"; [ $rc -ne 0 ] && exit 1
for dt in f32 fp16c; do for mode in one_phase sequential; do
  echo "== $dt, exchange $mode" | tee -a "$O/bench_group_ab.txt"
  if [ $mode = sequential ]; then export LUW_GROUP_EXCHANGE=sequential; else unset LUW_GROUP_EXCHANGE; fi
  timeout -k 10 900 python3 "$R/tools/bench_group.py" $dt quick 2>&1 | grep -v amdgpu.ids | tee -a "$O/bench_group_ab.txt"
done; done
