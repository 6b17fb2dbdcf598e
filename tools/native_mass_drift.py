#!/usr/bin/env python3
"""GPU box: drift of the mean density over K steps in a periodic box, FP16C DDFs, exact and native arithmetic (LUW_LIB picks the build), several seeds."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0,
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import latticeurbanwind_amd as luw
from helpers import synthetic_state
luw.load()
Nx, Ny, Nz = 256, 64, 64
for seed in (5, 6, 7):
    st = synthetic_state(Nx, Ny, Nz, seed=seed, solids=False, shell=None)
    for nat in (False, True):
        g = luw.LBM(Nx, Ny, Nz, 1e-4, fp16c=True, native_arith=nat)
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
        g.run(1); g.rho.read_from_device(); m0 = float(g.rho.data.astype(np.float64).mean())
        out = []
        for K in (200, 800):
            g.run(K); g.rho.read_from_device(); out.append(float(g.rho.data.astype(np.float64).mean()) - m0)
        g.close()
        print("seed %d %s: drift after 200 / 1000 steps %+.3e %+.3e" % (seed, "native" if nat else "exact ", out[0], out[1]), flush=True)
