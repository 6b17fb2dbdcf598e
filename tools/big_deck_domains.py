#!/usr/bin/env python3
"""A deck of the size of BASELINE configs[3] (2048x1024x512 cells, n_gpu = [4,2,1]) through `luw_driver` on the GPU box: all eight
domains in one process, here sharing the box's single GPU (--devices 0,0,..; 100 GB of lattice arrays).  'city' STL, VK inlet,
nudging, sponge, an unsteady output and an averaging window.  Prints the driver's phase timing (LUW_DRIVER_TIMING=1), the peak
resident memory of the process and sanity numbers of the written fields.
usage: big_deck_domains.py [fp32|fp16c] [Dx Dy Dz] [steps]"""
import glob, os, resource, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_refcases as mr
from vtkio import read_vtk
import numpy as np
ddf = sys.argv[1] if len(sys.argv) > 1 else "fp32"
D = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 2, 1)
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 24
tmp = tempfile.mkdtemp(dir=os.environ.get("LUW_BIG_TMP", "/tmp"))
s = 20.0          # geometry units -> metres
# 37 GB of output instead of 62: the box's disk holds 79
mr.write_case(tmp, "Tile", s, ["enable_buffer_nudging = true", "enable_top_sponge = true", "sponge_thickness_m = 64", "vk_inlet_l = 60", "vk_inlet_nmodes = 64",
    "n_gpu = [%d, %d, %d]" % D, "output_tke_ti_tls = []"],
              dims=(204.8, 102.4, 48.0), building="city", nstep=steps, unsteady=0, purge=8, vk=True, cell=0.1)
deck = os.path.join(tmp, "Tile", "conf.luwpf")
n = D[0] * D[1] * D[2]
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "latticeurbanwind_amd/host/luw_driver"), deck, "--ddf", ddf] + (["--devices", ",".join(["0"] * n)] if n > 1 else []),
                   capture_output=True, text=True, env=dict(os.environ, LUW_DRIVER_TIMING="1"))
wall = time.time() - t0
print("rc", r.returncode, "wall %.1f s, peak RSS of the driver %.1f GB" % (wall, resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1048576.0))
for l in r.stdout.splitlines():
    if any(k in l for k in ("Grid Resolution", "Domains", "halo faces", "Voxelized cells (whole", "profile boundaries mapped", "VK inlet", "Solver ",
            "Avg samples", "ERROR", "WARNING", "Error")):
        print(l[:200])
print(r.stderr[-3000:])
for f in sorted(glob.glob(os.path.join(tmp, "Tile", "RESULTS", "vtk", "*.vtk"))):
    if os.path.getsize(f) > (6 << 30):          # the 13 GB velocity files: header only
        print(os.path.basename(f), "%.1f GB" % (os.path.getsize(f) / 2 ** 30)); continue
    h, d = read_vtk(f)
    print(os.path.basename(f), h["dims"], {k: (float(np.nanmin(v)), float(np.nanmax(v)), bool(np.isfinite(v).all())) for k, v in d.items()})
import shutil; shutil.rmtree(tmp, ignore_errors=True)
