#!/usr/bin/env python3
"""Does the device address of the DDF array change the kernel time?  Creates several solvers of the same shape one after
the other (keeping some alive so that later ones land elsewhere), times each, prints the fi base address."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
from bench import channel_state
N = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 512, 512)
fl, u, rho = channel_state(*N)
keep = []
KEEP_ALL = os.environ.get("KEEP_ALL") == "1"
for k in range(int(os.environ.get("NOBJ", "8"))):
    g = luw.LBM(*N, 1.48e-7)
    g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
    g.run(0); g.run(10)
    t = sorted(g.run_timed(40) for _ in range(3))[1]
    p = g.device_ptr(capi.FIELD_FI)
    print("solver %d fi=0x%x (mod 2MiB=0x%x, mod 1GiB=0x%x) rho=0x%x  %.3f ms -> %.0f MLUPS" % (k, p, p % (2 << 20), p % (1 << 30),
        g.device_ptr(capi.FIELD_RHO), t, N[0] * N[1] * N[2] / t / 1e3))
    if KEEP_ALL or k % 2 == 0: keep.append(g)
    else: g.close()
for g in ([] if KEEP_ALL else keep):   # re-time the kept ones at the end
    t = sorted(g.run_timed(40) for _ in range(3))[1]
    print("kept fi=0x%x  %.3f ms" % (g.device_ptr(capi.FIELD_FI), t))
