#!/bin/bash
# GPU box: the one-process host without x boundary slabs (LUW_GROUP_X_SLABS=0) against the default, eight domains on the one
# GPU (tools/bench_group.py quick), FP32 and FP16C.   usage: tools/ab_group_xfree.sh <out dir>
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"; O="$1"; mkdir -p "$O"; : > "$O/bench_group_xfree.txt"
for dt in f32 fp16c; do for mode in no_x_slabs x_slabs; do
  echo "== $dt, $mode" | tee -a "$O/bench_group_xfree.txt"
  if [ $mode = x_slabs ]; then unset LUW_GROUP_X_SLABS; else export LUW_GROUP_X_SLABS=0; fi
  timeout -k 10 900 python3 "$R/tools/bench_group.py" $dt quick 2>&1 | grep -v amdgpu.ids | grep "4,2,1" | tee -a "$O/bench_group_xfree.txt"
done; done
