#!/usr/bin/env python3
"""Packs outputs of the REAL reference binary (runs made on the MI355X box with oracle/run_ref_case.sh, see
tests/golden/README.md) into small .npz fixtures:  u (SI units, as written by the reference) at t=8 and t=64,
rho at t=64, the TYPE_S mask recovered from the `fluid` field of the _avg VTK, plus the console log.
usage: pack_ref_outputs.py <run_dir> <out_prefix>   e.g.  gpurun_out/ref1/fp32_CaseA tests/golden/ref_fp32_CaseA
"""
import glob, os, re, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vtkio import read_vtk

run_dir, out = sys.argv[1], sys.argv[2]
pre = sys.argv[3] if len(sys.argv) > 3 else ""      # optional file-name prefix (dataset / multi-angle runs: DG_<u>_<angle>_, ANG_<a>_)
g = lambda pat: glob.glob(os.path.join(run_dir, pre + pat))[0]
d = {}
times = sorted(int(re.search(r"_raw_u-(\d+)\.vtk$", p).group(1)) for p in glob.glob(os.path.join(run_dir, pre + "*_raw_u-*.vtk")))
for t in (times[0], times[-1]):   # first unsteady output and the final step (8 and 64 for the 64-step cases)
    h, f = read_vtk(g("*_raw_u-%09d.vtk" % t)); d["u%d" % t] = f["data"].astype(np.float32)
h, f = read_vtk(g("*_raw_rho-*.vtk")); d["rho%d" % times[-1]] = f["data"][..., 0].astype(np.float32)
h, f = read_vtk(g("*_avg-*.vtk")); d["solid"] = (f["fluid"][..., 0] == 0)
if glob.glob(os.path.join(run_dir, pre + "*_raw_T-*.vtk")):       # temperature cases: final T in Kelvin and its window mean
    hT, fT = read_vtk(g("*_raw_T-*.vtk")); d["T%d" % times[-1]] = fT["data"][..., 0].astype(np.float32)
    if "T_avg" in f: d["T_avg"] = f["T_avg"][..., 0].astype(np.float32)
d["u_avg"] = f["u_avg"].astype(np.float32)   # mean of u over the last purge_avg=4 steps, SI units
d["dims"] = np.array(h["dims"]); d["origin"] = np.array(h["origin"]); d["spacing"] = np.array(h["spacing"])
np.savez_compressed(out + ".npz", **d)
log = open(os.path.join(run_dir, "console.log"), errors="replace").read()
keep = [l for l in log.splitlines() if l.strip() and not re.match(r"^\|\s+\d+\s+\|", l) and "MLUPs" not in l]
open(out + ".console.txt", "w").write("\n".join(keep) + "\n")
print(out, os.path.getsize(out + ".npz"))
