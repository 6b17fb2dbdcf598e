#!/usr/bin/env python3
"""Does the ORDER in which a wide lattice is swept matter?  One step of a 2048x512x128 lattice as one launch vs as x panels of
512 / 1024 cells launched one after the other (same kernel, same cells, same memory): times per step, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latticeurbanwind_amd as luw
from bench import channel_state
N = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 512, 128)
g = luw.LBM(*N, 1.48e-7)
fl, u, rho = channel_state(*N); g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
g.run(0); g.run(10)
st = torch.cuda.Stream(); g.set_stream(st.cuda_stream)
def sweep(panel, steps=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        for x0 in range(0, N[0], panel):
            g.enqueue_stream_collide((x0, min(x0 + panel, N[0]), 0, N[1], 0, N[2]), False)
        g.increment_time_step(1)
    e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / steps
res = {}
for rnd in range(5):
    for panel in (N[0], 1024, 512, 256):
        if panel <= N[0]: res.setdefault(panel, []).append(sweep(panel))
for panel, r in res.items():
    r = sorted(r); print("%s panel %4d: min %.3f median %.3f ms/step" % (N, panel, r[0], r[len(r) // 2]))
