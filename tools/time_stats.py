#!/usr/bin/env python3
"""Times the on-device Welford accumulation (k_stats_accumulate) on a 512x512x128 lattice: ms per sample and effective TB/s
(44 B read + 28 B written per cell)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latticeurbanwind_amd as luw
from bench import channel_state
N = (512, 512, 128)
g = luw.LBM(*N, 1.48e-7)
fl, u, rho = channel_state(*N)
g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
g.run(0); g.run(2)
g.stats_reset()
for _ in range(5): g.stats_accumulate()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.stats_accumulate()
g.finish(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
cells = N[0] * N[1] * N[2]
print("stats_accumulate: %.3f ms per sample = %.2f TB/s" % (dt * 1e3, cells * 72 / dt / 1e12))
