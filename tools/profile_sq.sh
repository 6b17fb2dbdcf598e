#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): shader-engine counters (instruction counts, VALU / memory-unit busy) for one bench.py
# configuration, one rocprofv3 --pmc pass per counter group.  usage: tools/profile_sq.sh <tag> <bench args...>
# Prints the per-launch averages for the stream_collide kernels; raw CSVs under gpurun_out/prof/<tag>/.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
TAG="$1"; shift
OUT="$R/gpurun_out/prof/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
# LUW_SQ_GROUPS="A B C;D E" replaces the counter groups (one pass each); "mix" = the dynamic instruction mix by class.
GROUPS_DEFAULT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM;VALUBusy SALUBusy;MemUnitBusy MemUnitStalled;WriteUnitStalled VALUUtilization;SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES;SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES"
GROUPS_MIX="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES;SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32;SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64;SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_VALU_FLOPS_FP32;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
G="${LUW_SQ_GROUPS:-$GROUPS_DEFAULT}"; [ "$G" = mix ] && G="$GROUPS_MIX"
IFS=';' read -ra GROUPS_ARR <<< "$G"
for c in "${GROUPS_ARR[@]}"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d "$OUT/sq_$n" -- python3 "$R/bench.py" --no-cpu-baseline --steps 20 --warmup 3 "$@" > "$OUT/bench_sq_$n.json" 2> "$OUT/sq_$n.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/sq_*/*/*_counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_stream_collide" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, name), v in per.items():
        acc[name][0] += v; acc[name][1] += 1
for name in sorted(acc):
    print("%-22s %16.1f  (mean over %d launches)" % (name, acc[name][0] / acc[name][1], acc[name][1]))
PY
