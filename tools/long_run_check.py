#!/usr/bin/env python3
"""Long-run sanity of a bench workload on the GPU: STEPS steps (default 3000) of the 1024x1024x256 building case, then rho / u finite everywhere,
mean density of the fluid cells against the start, max |u|.  usage: long_run_check.py [f32|fp16c] [steps] [c2|c3] [exact|native]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import latticeurbanwind_amd as luw
from bench import channel_state, WORKLOADS
dt = sys.argv[1] if len(sys.argv) > 1 else "fp16c"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
arith = sys.argv[4] if len(sys.argv) > 4 else "exact"
N, bld, _ = WORKLOADS[wl]
g = luw.LBM(*N, 1.48e-7, fp16c=(dt == "fp16c"), native_arith=(arith == "native"))
fl, u, rho = channel_state(*N, buildings=bld)
g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
fluid = (fl & 0x03) == 0
m0 = float(rho[fluid].astype(np.float64).mean())
t0 = time.time(); g.run(steps); g.u.read_from_device(); g.rho.read_from_device(); dtw = time.time() - t0
U = g.u.data.reshape(3, -1)
ok = bool(np.isfinite(g.rho.data).all() and np.isfinite(g.u.data).all())
m1 = float(g.rho.data[fluid].astype(np.float64).mean())
print("%s %s %s %d steps in %.1f s: finite %s, mean rho of fluid cells %.8f -> %.8f, max |u| %.4f (inflow 0.1)" % (wl, dt, arith, steps, dtw, ok, m0, m1,
    float(np.abs(U).max())))
assert ok and abs(m1 - m0) < 5e-3 and float(np.abs(U).max()) < 0.45
g.close()
