cd /tmp && export TMPDIR=/tmp
for cut in "4 1 1" "1 4 2"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/prof/xhalo_$(echo $cut | tr ' ' '_'); mkdir -p $d
  PROBE_REPS=6 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/box_rate_probe.py f32 512 512 512 $cut > $d/out.txt 2> $d/err.txt
  PROBE_REPS=6 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d ${d}_hit -- python3 $GRAFT_REPO_ROOT/tools/box_rate_probe.py f32 512 512 512 $cut > ${d}_hit/out.txt 2> ${d}_hit/err.txt
  python3 - $d ${d}_hit "$cut" <<'PY'
import csv, glob, sys
for dd in sys.argv[1:3]:
    f = glob.glob(dd + "/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_stream_collide" in r["Kernel_Name"]]
    by = {}
    for r in rows: by.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0); by[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(by, key=int)[2:8]          # the six timed launches of the first box (whole), behind two warm-up launches
    names = sorted(by[ids[0]])
    print("cut", sys.argv[3], "whole box, mean of", len(ids), "launches:", {n: round(sum(by[i][n] for i in ids) / len(ids)) for n in names})
PY
done
