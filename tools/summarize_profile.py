#!/usr/bin/env python3
"""Condenses one tools/profile_bench.sh run (gpurun_out/prof/<tag>/) into profiles/<tag>_summary.json + <tag>_kernel_stats.csv.
HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950: read requests are 128 B
(FETCH_SIZE counts 64-byte units as KB -> x2; cross-checked with TCC_EA0_RDREQ x 128 B), WRITE_SIZE is in KB.
usage: summarize_profile.py <tag> <kernel-name-substring> [algorithmic bytes per launch; default: what bench.py reported in the traced run]"""
import csv, glob, json, os, shutil, sys

tag, kname = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prof", tag)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")

bench = json.loads(open(os.path.join(root, "bench_trace.json")).read().strip().splitlines()[-1])
algo = float(sys.argv[3]) if len(sys.argv) > 3 else float(bench["roofline"]["algorithmic_bytes_per_launch"])   # (N > 1-style lines carry none: pass it)

def one(pattern):   # the newest match: gpurun_out/ keeps the files of earlier runs of the same tag
    return max(glob.glob(os.path.join(root, pattern)), key=os.path.getmtime)

stats = one("trace/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))
calls = tot = 0
for r in csv.DictReader(open(stats)):
    if kname in r["Name"]:
        calls += int(r["Calls"]); tot += float(r["TotalDurationNs"])
def counter(dirpat, name):
    vals = {}
    for r in csv.DictReader(open(one(dirpat + "/*/*_counter_collection.csv"))):
        if kname in r["Kernel_Name"] and r["Counter_Name"] == name:
            vals.setdefault(r["Dispatch_Id"], 0.0); vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return sum(vals.values()) / max(len(vals), 1)
fetch_kb = counter("pmc_FETCH_SIZE", "FETCH_SIZE"); write_kb = counter("pmc_WRITE_SIZE", "WRITE_SIZE")
rd = counter("pmc_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum", "TCC_EA0_RDREQ_sum"); wr = counter("pmc_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_sum")
hit = counter("pmc_TCC_HIT_sum_TCC_MISS_sum", "TCC_HIT_sum"); miss = counter("pmc_TCC_HIT_sum_TCC_MISS_sum", "TCC_MISS_sum")
def optional(dirpat, name):
    try: return counter(dirpat, name)
    except ValueError: return None               # pass not collected (older runs)
valu_busy, mem_stalled = optional("pmc_VALUBusy_MemUnitStalled", "VALUBusy"), optional("pmc_VALUBusy_MemUnitStalled", "MemUnitStalled")
waves, insts_valu, insts_salu = (optional("pmc_SQ_WAVES_SQ_INSTS_VALU_SQ_INSTS_SALU", n) for n in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"))
traffic = rd * 128.0 + write_kb * 1024.0
s = {"tag": tag, "workload": bench["config"]["workload"], "rows_per_xcd": bench["config"].get("rows_per_xcd"),
    "arith": bench["config"].get("arith"), "kernel": kname, "launches": calls, "rocprof_avg_ms": round(tot / calls / 1e6, 4),
    "hip_event_avg_ms_same_run": bench["roofline"]["kernel_ms"],
     "algorithmic_bytes_per_launch": algo, "FETCH_SIZE_KB": fetch_kb, "FETCH_bytes_corrected_x2": fetch_kb * 1024.0 * 2.0, "TCC_EA0_RDREQ": rd,
     "read_bytes_128B_requests": rd * 128.0, "WRITE_SIZE_KB": write_kb, "write_bytes": write_kb * 1024.0, "TCC_EA0_WRREQ": wr,
     "hbm_traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": round(traffic / algo, 3), "L2_hit_rate": round(hit / max(hit + miss, 1.0), 4),
     "achieved_GBps_algorithmic": round(algo / (tot / calls) , 1),
     "VALUBusy_percent": None if valu_busy is None else round(valu_busy, 1), "MemUnitStalled_percent": None if mem_stalled is None else round(mem_stalled, 2),
     "waves_per_launch": waves, "valu_insts_per_wave": None if not waves else round(insts_valu / waves, 1), "salu_insts_per_wave": None if not waves
         else round(insts_salu / waves, 1)}
json.dump(s, open(os.path.join(out, tag + "_summary.json"), "w"), indent=1)
print(json.dumps(s))
