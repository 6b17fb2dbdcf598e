#!/usr/bin/env python3
"""One rank's domain (loopback halos), SAME solver object: alternates shell/interior overlap on and off, several rounds, so that
the physical placement of the arrays (which differs between objects) cancels.  usage: ab_overlap.py Dx Dy Dz Nx Ny Nz [f32|fp16c]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latticeurbanwind_amd as luw
from latticeurbanwind_amd.distributed import DomainDecomposedLBM
from bench import channel_state
from tools.bench_domain_overhead import Loopback
D = tuple(int(v) for v in sys.argv[1:4]); size = tuple(int(v) for v in sys.argv[4:7]); dt = sys.argv[7] if len(sys.argv) > 7 else "f32"
luw.load()
N = tuple(s * d for s, d in zip(size, D))
sim = DomainDecomposedLBM(N, D, 1.48e-7, rank=0, transport=Loopback(), overlap=True, device=0, fp16c=(dt == "fp16c"))
fl, u, rho = channel_state(sim.lNx, sim.lNy, sim.lNz, *sim.global_offset, *N)
sim.set_fields(fl, u, rho); sim.initialize(); sim.run(10)
res = {True: [], False: []}
for rnd in range(5):
    for ov in (True, False):
        sim.overlap = ov
        torch.cuda.synchronize(); t0 = time.perf_counter(); sim.run(40); torch.cuda.synchronize()
        res[ov].append((time.perf_counter() - t0) / 40 * 1e3)
for ov in (True, False):
    print("D=%s local=%s %s overlap=%-5s ms/step: %s" % (D, (sim.lNx, sim.lNy, sim.lNz), dt, ov, " ".join("%.3f" % v for v in res[ov])))
