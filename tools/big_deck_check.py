#!/usr/bin/env python3
"""Scale check of the deck driver (GPU box): a 1024x1024x256 profile deck with the 'city' STL (scaled up), VK inlet, nudging,
sponge, unsteady outputs and averaging; prints the driver's timing rows and sanity numbers of the written fields."""
import glob, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_refcases as mr
from vtkio import read_vtk
import numpy as np
tmp = tempfile.mkdtemp()
s = 20.0          # geometry units -> metres: the city spans 1920 x 1600 m
mr.write_case(tmp, "Big", s, ["enable_buffer_nudging = true", "enable_top_sponge = true", "sponge_thickness_m = 64", "vk_inlet_l = 60", "vk_inlet_nmodes = 64"],
              dims=(102.4, 102.4, 22.4), building="city", nstep=40, unsteady=20, purge=8, vk=True, cell=0.1)
deck = os.path.join(tmp, "Big", "conf.luwpf")
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "latticeurbanwind_amd/host/luw_driver"), deck, "--ddf", sys.argv[1] if len(sys.argv) > 1 else "fp16c"],
    capture_output=True, text=True)
print("rc", r.returncode, "wall %.1f s" % (time.time() - t0))
for l in r.stdout.splitlines():
    if any(k in l
            for k in ("Grid Resolution", "Voxelized cells (whole", "profile boundaries mapped", "VK inlet", "Solver ", "Avg samples", "ERROR", "WARNING")):
        print(l)
for f in sorted(glob.glob(os.path.join(tmp, "Big", "RESULTS", "vtk", "*.vtk"))):
    h, d = read_vtk(f)
    print(os.path.basename(f), h["dims"], {k: (float(np.nanmin(v)), float(np.nanmax(v)), bool(np.isfinite(v).all())) for k, v in d.items()})
