#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace of one rank-shape block of bench.py: what each kernel of the decomposed step costs per step (halo pack / unpack / edge kernels
# next to the step kernels).   usage: tools/profile_rank_shape.sh <block> <rccl-self|peer-loopback> [steps]      -> gpurun_out/prof/rank_<block>_<transport>.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
BLK="$1"; TR="$2"; STEPS="${3:-100}"
OUT="$R/gpurun_out/prof/rank_${BLK}_${TR}"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --rank-shape-block "$BLK" --rank-transport "$TR" --steps "$STEPS" --warmup 10 > "$OUT/bench.json" 2> "$OUT/trace.err"
python3 - "$OUT" "$STEPS" <<'PY' > "$OUT.txt"
import csv, glob, json, os, sys
out, steps = sys.argv[1], int(sys.argv[2]) + 10
stats = max(glob.glob(os.path.join(out, "trace/*/*_kernel_stats.csv")), key=os.path.getmtime)
line = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
print("# %s: %s ms/step (interior kernel %s, shell %s, exchange %s)" % (os.path.basename(out), line.get("ms_per_step"), line.get("kernel_ms"), line.get("shell_ms"), line.get("exchange_ms")))
print("# kernel, calls, calls per step, average us, total ms per step")
for r in csv.DictReader(open(stats)):
    calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
    if calls >= steps // 2:
        print("%-90s %6d %5.1f %9.1f %8.4f" % (r["Name"][:90], calls, calls / steps, tot / calls / 1e3, tot / steps / 1e6))
PY
python3 - "$OUT" <<'PY' >> "$OUT.txt"
import csv, glob, os, re, sys
out = sys.argv[1]
tr = max(glob.glob(os.path.join(out, "trace/*/*_kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2: len(rows) // 2 + 44]                 # a window in the middle of the run: two or three steps
t0 = int(rows[0]["Start_Timestamp"])
print("# timeline (us from the first row): start, duration, queue, kernel [grid]")
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70]
    print("%9.1f %8.1f  q%-3s %s [%s x %s x %s]" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        r.get("Queue_Id", "?"), name, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?")))
PY
cat "$OUT.txt"
