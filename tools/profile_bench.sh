#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + HBM counters for one bench.py configuration.
# usage: tools/profile_bench.sh <tag> <bench args...>     outputs under gpurun_out/prof/<tag>/
# Counters are collected in their own passes (gfx950: FETCH_SIZE and WRITE_SIZE do not fit one pass;
# --pmc must not be combined with trace domains other than --kernel-trace).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
TAG="$1"; shift
OUT="$R/gpurun_out/prof/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --no-cpu-baseline --no-secondary "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "VALUBusy MemUnitStalled" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$n" -- python3 "$R/bench.py" --no-cpu-baseline --no-secondary "$@" > "$OUT/bench_pmc_$n.json" 2> "$OUT/pmc_$n.err"
done
find "$OUT" -name "*.csv" | head -30
