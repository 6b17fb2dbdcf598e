#!/usr/bin/env python3
"""Step time of one solver over wall time from a cold box: batches of 100 steps for SECONDS (default 60), mean kernel ms per batch.
usage: warm_drift.py [f32|fp16c] [seconds] [Nx Ny Nz]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeurbanwind_amd as luw
from bench import channel_state
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
N = tuple(int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (512, 512, 512)
g = luw.LBM(*N, 1.48e-7, fp16c=(dt == "fp16c"))
fl, u, rho = channel_state(*N)
g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
g.run(0)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    ms = g.run_timed(100); n += 100
    print("t = %6.2f s  steps %6d  kernel %.4f ms" % (time.time() - t0, n, ms), flush=True)
    if "--idle" in sys.argv and n % 1000 == 0:
        time.sleep(3.0)
g.close()
