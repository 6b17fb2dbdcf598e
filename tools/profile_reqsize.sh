#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): HBM read / write requests of the step kernel BY SIZE (TCC_EA0_RDREQ = all, _32B, _64B, _128B; WRREQ = all, _64B), so
# that the traffic is counted exactly instead of as "requests x 128 B".  usage: tools/profile_reqsize.sh <tag> <bench args...>
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
TAG="$1"; shift
OUT="$R/gpurun_out/reqsize/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
i=0
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -- python3 "$R/bench.py" --no-cpu-baseline --no-secondary --steps 40 --warmup 5 "$@" > "$OUT/bench_pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed: $(tail -2 $OUT/pass$i.err)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
tot = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(sys.argv[1] + "/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stream_collide" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1].add(r["Dispatch_Id"])
v = {k: tot[k][0] / max(len(tot[k][1]), 1) for k in tot}
for k in sorted(v): print("%-28s %16.0f per launch" % (k, v[k]))
b = json.loads(open(sys.argv[1] + "/bench_pass1.json").read().strip().splitlines()[-1])
algo = b["roofline"]["algorithmic_bytes_per_launch"]
g = lambda n: v.get(n, 0.0)
rd = 32 * g("TCC_EA0_RDREQ_32B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum")
other = g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum") - g("TCC_EA0_RDREQ_64B_sum") - g("TCC_EA0_RDREQ_128B_sum")
print("read bytes by size %.3f GB (+ %d requests of no listed size); as requests x 128 B %.3f GB; algorithmic bytes (read + write) %.3f GB" % (rd / 1e9, other, 128 * g("TCC_EA0_RDREQ_sum") / 1e9, algo / 1e9))
print("write bytes %.3f GB (64-byte requests %.0f of %.0f)" % ((64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum"))) / 1e9, g("TCC_EA0_WRREQ_64B_sum"), g("TCC_EA0_WRREQ_sum")))
PY
