#!/usr/bin/env python3
"""Long, large cross-check of the two FP16C product kernels on the GPU box: the same 512x512x128 channel with the building
array, Coriolis on, for STEPS steps on the pair kernel and on the scalar kernel; u, rho and all 19 DDF planes must be identical
bit for bit.  usage: check_pair_vs_scalar.py [STEPS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
from bench import channel_state, coriolis_omega
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = (512, 512, 128)
fl, u, rho = channel_state(*N, buildings=True)
res = []
for kern in (capi.KERNEL_PAIR, capi.KERNEL_SCALAR):
    g = luw.LBM(*N, 1.48e-7, fp16c=True, kernel=kern)
    g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
    g.set_coriolis(*coriolis_omega())
    g.run(0); g.run(steps)
    g.u.read_from_device(); g.rho.read_from_device()
    res.append((g.u.data.copy(), g.rho.data.copy(), np.asarray(g.download_fi()).copy()))
    g.close()
for name, a, b in zip(("u", "rho", "fi"), res[0], res[1]):
    same = np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    print("%-4s identical: %s" % (name, same))
    assert same
print("pair == scalar after %d steps on %s (u range %.4f..%.4f)" % (steps, N, float(res[0][0].min()), float(res[0][0].max())))
