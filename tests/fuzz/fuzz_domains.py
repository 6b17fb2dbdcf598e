#!/usr/bin/env python3
"""Random decompositions on one GPU (LocalGroup: several HIP domains in lock-step, real pack/unpack kernels, shell/interior split)
against the single-domain CPU oracle: random domain grids, lattice extents (odd and even rows), overlap on/off, FP32 / FP16C,
scalar / pair kernel, sometimes with the thermal lattice.  usage: fuzz_domains.py [CASES] [SEED]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
from latticeurbanwind_amd.distributed import LocalGroup, HipDomain
from oracle import oracle
from helpers import synthetic_state, thermal_state

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
GRIDS = [(2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 1), (1, 2, 2), (2, 1, 2), (2, 2, 2), (1, 4, 2), (1, 3, 2), (4, 2, 1), (3, 1, 1), (1, 2, 4)]
luw.load()
for k in range(cases):
    D = GRIDS[int(rng.integers(0, len(GRIDS)))]
    per = (int(rng.integers(3, 40)), int(rng.integers(3, 9)), int(rng.integers(3, 8)))
    if rng.integers(0, 3) == 0 and D[0] == 1:
        per = (int(rng.integers(250, 300)), per[1], per[2])          # rows wide enough for the automatic pair kernel
    gN = tuple(p * d for p, d in zip(per, D))
    fp16c = bool(rng.integers(0, 2)); overlap = bool(rng.integers(0, 2)); thermal = rng.integers(0, 4) == 0
    kern = capi.KERNEL_PAIR if (fp16c and rng.integers(0, 2)) else capi.KERNEL_AUTO
    flags, u, rho = synthetic_state(*gN, seed=int(rng.integers(0, 1 << 30)), shell=[None, "luw"][int(rng.integers(0, 2))])
    steps = int(rng.integers(2, 7))
    kw = dict(alpha=0.004) if thermal else {}
    grp = LocalGroup(gN, D, 0.01, lambda lay: HipDomain(lay, 0.01, fp16c=fp16c, kernel=kern, **kw), overlap=overlap)
    o = oracle.OracleLBM(*gN, 0.01, fp16c=fp16c, **kw)
    if thermal:
        tflags, T = thermal_state(flags, gN)
        for s in grp.sims: s.set_fields_from_global(tflags, u, rho, T)
        o.flags[:] = tflags; o.u[:] = u; o.rho[:] = rho; o.T[:] = T
    else:
        for s in grp.sims: s.set_fields_from_global(flags, u, rho)
        o.flags[:] = flags; o.u[:] = u; o.rho[:] = rho
    grp.run(steps); o.run(steps)
    gu, grho = grp.gather_u_rho()
    ok = np.array_equal(gu, o.u) and np.array_equal(grho, o.rho)
    if thermal:
        gT = np.zeros((1, gN[2], gN[1], gN[0]), np.float32)
        for s in grp.sims:
            tb, off = s.interior_to_global(s.backend.download_T(), 1)
            gT[:, off[2]:off[2] + tb.shape[1], off[1]:off[1] + tb.shape[2], off[0]:off[0] + tb.shape[3]] = tb
        ok = ok and np.array_equal(gT.ravel(), o.T)
    print("%3d  lattice %-14s n_gpu %-9s %s kernel %d overlap %d thermal %d steps %d : %s" % (k, gN, D, "fp16c" if fp16c else "f32  ", kern, overlap, thermal,
        steps, "ok" if ok else "MISMATCH"), flush=True)
    for s in grp.sims: s.backend.lbm.close()
    assert ok
print("all %d decomposed cases identical to the single-domain oracle" % cases)
