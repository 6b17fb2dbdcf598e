"""One fresh process: a solver large enough for the placement search of its DDF array (luw_create, tune_ddf_placement), every candidate tried
(LUW_TUNE_FAST=99 in the environment), then stepped against the CPU oracle; a second large solver in the same process (no second search)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import latticeurbanwind_amd as luw
from oracle import oracle
from helpers import synthetic_state

fp16c = len(sys.argv) > 1 and sys.argv[1] == "fp16c"
Nx, Ny, Nz = (512, 256, 512) if fp16c else (512, 256, 256)         # 19 planes of 64 / 32 M cells: 2.5 GB either way
st = synthetic_state(Nx, Ny, Nz, seed=5, shell="luw")
import torch
from latticeurbanwind_amd import capi
if len(sys.argv) > 2 and sys.argv[2] == "replace":       # the search sees the array in place 30 % slower than it is: another draw has to take its place
    capi.inject_fault(capi.FAULT_SLOW_FIRST_PLACEMENT)
for cycle in range(2):
    free0 = torch.cuda.mem_get_info(0)[0]
    g = luw.LBM(Nx, Ny, Nz, 1e-4, fp16c=fp16c)
    print("placement", capi.placement_info(g._h), flush=True)
    print("device memory used by the solver %.2f GB" % ((free0 - torch.cuda.mem_get_info(0)[0]) / 1e9), flush=True)
    o = oracle.OracleLBM(Nx, Ny, Nz, 1e-4, fp16c=fp16c)
    g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    g.run(3); o.run(3)
    g.u.read_from_device(); g.rho.read_from_device()
    a = np.asarray(g.download_fi()).copy(); b = np.asarray(o.fi).copy()
    if fp16c:
        a[a == 0x8000] = 0; b[b == 0x8000] = 0
    print("cycle %d equal %s" % (cycle, bool(np.array_equal(g.u.data, o.u) and np.array_equal(g.rho.data, o.rho) and np.array_equal(a, b))), flush=True)
    g.close()
