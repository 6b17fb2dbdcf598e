#!/usr/bin/env bash
# GPU box: per-channel L2 -> memory request counters of the step kernel for one bench.py configuration under several allocation policies of the DDF array
# (LUW_ALLOC; placement search off): TCC_EA0_RDREQ / TCC_EA0_WRREQ / TCC_BUSY come with dimensions INSTANCE[0:15] x XCC[0:7] = 128 channels.  Does a
# slow-class placement show as an imbalance between the channels?   usage: tools/profile_channels.sh <tag> "<alloc> <alloc> ..." <bench args...>
# -> gpurun_out/r04_channels_<tag>.txt: per policy the kernel time and, per counter, mean / min / max over the 128 channels and the coefficient of variation
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
TAG="$1"; ALLOCS="$2"; shift 2
mkdir -p "$R/gpurun_out/prof"
OUT="$R/gpurun_out/r04_channels_$TAG.txt"; : > "$OUT"
cd /tmp; export TMPDIR=/tmp
for alloc in $ALLOCS; do
  for c in TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_BUSY; do
    D="$R/gpurun_out/prof/chan_${TAG}_$(echo $alloc | tr ':' '_')_$c"
    LUW_ALLOC=$alloc LUW_TUNE_PLACEMENT=0 rocprofv3 --pmc $c --output-format json -d "$D" -- python3 "$R/bench.py" --no-cpu-baseline --no-secondary --steps 20 --warmup 5 "$@" > "$D.json" 2> "$D.err" || echo "pass failed ($alloc $c): $(tail -1 $D.err)" >> "$OUT"
    python3 - "$D" "$alloc" "$D.json" "$c" >> "$OUT" <<'PY'
import glob, sys, json, math
# (rocprofv3's CSV output sums the 128 instances of a TCC counter; the JSON output keeps one record per instance and dispatch)
d = json.load(open(glob.glob(sys.argv[1] + "/*/*results.json")[0]))["rocprofiler-sdk-tool"][0]
names = {k["kernel_id"]: k["kernel_name"] for k in d["kernel_symbols"]}
per = None; n = 0
for e in d["callback_records"]["counter_collection"]:
    if "k_stream_collide" not in names.get(e["dispatch_data"]["dispatch_info"]["kernel_id"], ""): continue
    v = [r["value"] for r in e["records"]]
    per = v if per is None else [a + b for a, b in zip(per, v)]; n += 1
try: ms = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["roofline"]["kernel_ms"]
except Exception: ms = None
if not per: print("%s %s: no records" % (sys.argv[2], sys.argv[4]))
else:
    v = [x / n for x in per]; m = sum(v) / len(v); s = sorted(v)
    xcc = [sum(v[i * 16:(i + 1) * 16]) / 16 for i in range(len(v) // 16)]
    print("%-9s kernel %s ms  %-14s channels %3d  mean %.5g  min %.5g  max %.5g  max/mean %.4f  cv %.5f  | per-XCD means / overall: %s" % (sys.argv[2], ms, sys.argv[4], len(v), m,
        s[0], s[-1], s[-1] / m, math.sqrt(sum((x - m) ** 2 for x in v) / len(v)) / m, " ".join("%.3f" % (x / m) for x in xcc)))
PY
  done
done
cat "$OUT"
