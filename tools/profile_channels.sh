#!/usr/bin/env bash
# GPU box: per-channel L2 -> memory request counters (TCC_EA0_RDREQ[i], TCC_EA0_WRREQ[i], TCC_BUSY[i]) of the step kernel for one bench.py configuration
# under several allocation policies of the DDF array (LUW_ALLOC), placement search off: does a slow-class placement show as channel imbalance?
# usage: tools/profile_channels.sh <tag> "<alloc> <alloc> ..." <bench args...>     -> gpurun_out/r04_channels_<tag>.txt
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
TAG="$1"; ALLOCS="$2"; shift 2
OUT="$R/gpurun_out/r04_channels_$TAG.txt"; : > "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RDREQ\[[0-9]*\]\|TCC_EA0_WRREQ\[[0-9]*\]\|TCC_BUSY\[[0-9]*\]\|TCC_EA0_RDREQ_sum\|TCC_REQ\[[0-9]*\]" | sort -u | tr '\n' ' ' >> "$OUT"; echo >> "$OUT"
for alloc in $ALLOCS; do
  for grp in "TCC_EA0_RDREQ[0] TCC_EA0_RDREQ[1] TCC_EA0_RDREQ[2] TCC_EA0_RDREQ[3] TCC_EA0_RDREQ[4] TCC_EA0_RDREQ[5] TCC_EA0_RDREQ[6] TCC_EA0_RDREQ[7]" \
             "TCC_EA0_RDREQ[8] TCC_EA0_RDREQ[9] TCC_EA0_RDREQ[10] TCC_EA0_RDREQ[11] TCC_EA0_RDREQ[12] TCC_EA0_RDREQ[13] TCC_EA0_RDREQ[14] TCC_EA0_RDREQ[15]" \
             "TCC_BUSY[0] TCC_BUSY[1] TCC_BUSY[2] TCC_BUSY[3] TCC_BUSY[4] TCC_BUSY[5] TCC_BUSY[6] TCC_BUSY[7]" \
             "TCC_BUSY[8] TCC_BUSY[9] TCC_BUSY[10] TCC_BUSY[11] TCC_BUSY[12] TCC_BUSY[13] TCC_BUSY[14] TCC_BUSY[15]"; do
    D="$R/gpurun_out/prof/chan_${TAG}_$(echo $alloc$grp | md5sum | cut -c1-8)"
    LUW_ALLOC=$alloc LUW_TUNE_PLACEMENT=0 rocprofv3 --pmc $grp --output-format csv -d "$D" -- python3 "$R/bench.py" --no-cpu-baseline --no-secondary --steps 20 --warmup 5 "$@" > "$D.json" 2> "$D.err" || echo "pass failed ($alloc): $(tail -1 $D.err)" >> "$OUT"
    python3 - "$D" "$alloc" "$D.json" >> "$OUT" <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stream_collide" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
try: ms = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["roofline"]["kernel_ms"]
except Exception: ms = None
vals = {k: v[0] / max(len(v[1]), 1) for k, v in acc.items()}
if vals:
    m = sum(vals.values()) / len(vals)
    print("%-10s kernel %s ms  " % (sys.argv[2], ms) + "  ".join("%s %.4g" % (k, vals[k]) for k in sorted(vals, key=lambda s: int(s.split("[")[1][:-1]) if "[" in s else 0)) + "   max/mean %.3f" % (max(vals.values()) / m if m else 0))
PY
  done
done
cat "$OUT"
