#!/usr/bin/env python3
"""GPU box: one rank of n_gpu = [4,2,1] (local 514x514x512) in the reference's SHIPPED configuration -- FP16C DDFs and the thermal D3Q7 lattice -- stepped by
the production host with the peer-loopback buffers; ms per step.  Run it under LUW_X_FACE_FUSED=0 / LUW_X_INSERT_FUSED=0 for the pack / unpack kernels instead
of the x faces written and read by the step kernels.   usage: rank_thermal_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latticeurbanwind_amd as luw
from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, PeerLoopbackTransport
from bench import fill_channel, NU

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
luw.load()
D, gN = (4, 2, 1), (2048, 1024, 512)
lay = DomainLayout(gN, D, 0)
sim = DomainDecomposedLBM(gN, D, NU, rank=0, transport=PeerLoopbackTransport(lay), fp16c=True, device=0, alpha=2.1e-7)
lb = sim.backend.lbm
fill_channel(lb.flags.data, lb.u.data, lb.rho.data, *lay.lN, *lay.O, *gN, buildings=True)
lb.T.data[:] = 1.0
sim.initialize(); sim.run(20)
torch.cuda.synchronize(); t0 = time.perf_counter(); sim.run(steps); torch.cuda.synchronize()
print("thermal FP16C rank of [4,2,1], x faces out %s in %s: %.4f ms/step" % (os.environ.get("LUW_X_FACE_FUSED", "1"), os.environ.get("LUW_X_INSERT_FUSED", "1"),
    (time.perf_counter() - t0) / steps * 1e3))
sim.backend.close()
