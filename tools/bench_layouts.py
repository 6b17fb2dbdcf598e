#!/usr/bin/env python3
"""Interleaved A/B of one rank's step time for several decompositions on ONE GPU (loopback halo transport, see
bench_domain_overhead.py).  All candidates are set up first, then timed round-robin several times so that clock / thermal
drift of the device hits all of them alike; prints min and median ms/step per candidate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import latticeurbanwind_amd as luw
from latticeurbanwind_amd.distributed import DomainDecomposedLBM
from bench import channel_state
from tools.bench_domain_overhead import Loopback

CANDS = [  # (label, D, per-GPU size, overlap)
    ("single 512^3", (1, 1, 1), (512, 512, 512), True),
    ("[4,2,1] 512^3 sequential", (4, 2, 1), (512, 512, 512), False),
    ("[4,2,1] 512^3 overlap", (4, 2, 1), (512, 512, 512), True),
    ("[1,4,2] 512^3 overlap", (1, 4, 2), (512, 512, 512), True),
    ("[1,2,4] 2048x512x128 overlap", (1, 2, 4), (2048, 512, 128), True),
    ("[1,2,4] 2048x512x128 sequential", (1, 2, 4), (2048, 512, 128), False),
    ("[1,4,2] 2048x256x256 overlap", (1, 4, 2), (2048, 256, 256), True),
    ("[1,2,2] 1024x512x256 overlap", (1, 2, 2), (1024, 512, 256), True),
    ("[1,4,1] 1024x256x512 overlap", (1, 4, 1), (1024, 256, 512), True),
    ("[1,2,1] 1024x256x512 overlap", (1, 2, 1), (1024, 256, 512), True),
]
if len(sys.argv) > 1:
    CANDS = [c for c in CANDS if any(k in c[0] for k in sys.argv[1:])]
luw.load()
sims = []
for label, D, size, ov in CANDS:
    N = tuple(s * d for s, d in zip(size, D))
    sim = DomainDecomposedLBM(N, D, 1.48e-7, rank=0, transport=Loopback(), overlap=ov, device=0)
    fl, u, rho = channel_state(sim.lNx, sim.lNy, sim.lNz, *sim.global_offset, *N)
    sim.set_fields(fl, u, rho); sim.initialize(); sim.run(5)
    sims.append(sim)
torch.cuda.synchronize()
res = {c[0]: [] for c in CANDS}
for rnd in range(6):
    for (label, D, size, ov), sim in zip(CANDS, sims):
        torch.cuda.synchronize(); t0 = time.perf_counter(); sim.run(40); torch.cuda.synchronize()
        res[label].append((time.perf_counter() - t0) / 40 * 1e3)
for label, D, size, ov in CANDS:
    r = sorted(res[label]); cells = size[0] * size[1] * size[2]
    print("%-34s min %.3f  median %.3f ms/step -> %.0f MLUPS/GPU (median)" % (label, r[0], r[len(r) // 2], cells / (r[len(r) // 2] * 1e-3) / 1e6))
