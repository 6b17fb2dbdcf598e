#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): where the waves of one bench.py configuration spend their cycles -- SQ wait / active counters of the dominant
# kernel in their own rocprofv3 passes (8 SQ slots per pass on gfx950).  usage: tools/profile_stalls.sh <tag> <bench args...>
# outputs under gpurun_out/stalls/<tag>/; a line per counter (mean over the dispatches of the step kernels) on stdout.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
TAG="$1"; shift
OUT="$R/gpurun_out/stalls/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
         "SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -- python3 "$R/bench.py" --no-cpu-baseline --no-secondary --steps 40 --warmup 5 "$@" > "$OUT/bench_pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed: $(tail -2 $OUT/pass$i.err)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(sys.argv[1] + "/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stream_collide" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1].add(r["Dispatch_Id"])
for k in sorted(tot):
    print("%-28s %16.0f per launch" % (k, tot[k][0] / max(len(tot[k][1]), 1)))
PY
