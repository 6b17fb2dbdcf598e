#!/usr/bin/env python3
"""Random-shape fuzz of the product kernels against the CPU oracle (GPU box): lattices with random extents (odd and even row
lengths, rows shorter and longer than a block), random solid / TYPE_E cells, random forces; FP16C on the pair kernel and FP32 in
both addressing forms.  Any mismatch in u, rho or a DDF plane stops the run.  usage: fuzz_kernels.py [CASES] [SEED]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
from oracle import oracle
from helpers import synthetic_state

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for k in range(cases):
    Nx = int(rng.choice([rng.integers(2, 40), rng.integers(120, 140), rng.integers(250, 300), rng.integers(500, 530)]))
    Ny, Nz = int(rng.integers(2, 12)), int(rng.integers(2, 9))
    fp16c = bool(rng.integers(0, 2))
    kern = capi.KERNEL_PAIR if fp16c and rng.integers(0, 4) else capi.KERNEL_SCALAR
    shell = [None, "luw"][int(rng.integers(0, 2))]
    st = synthetic_state(Nx, Ny, Nz, seed=int(rng.integers(0, 1 << 30)), shell=shell, solids=True)
    force = tuple(float(v) for v in (rng.standard_normal(3) * 1e-5)) if rng.integers(0, 2) else (0.0, 0.0, 0.0)
    cor = (0.0, 3e-5, 4e-5) if rng.integers(0, 3) == 0 else None
    g = luw.LBM(Nx, Ny, Nz, 1e-3, *force, fp16c=fp16c, kernel=kern, update_fields_every_step=True)
    o = oracle.OracleLBM(Nx, Ny, Nz, 1e-3, *force, fp16c=fp16c)
    for l in (g, o):
        (l.flags.data if hasattr(l.flags, "data") else l.flags)[:] = st[0]
        (l.u.data if hasattr(l.u, "data") else l.u)[:] = st[1]
        (l.rho.data if hasattr(l.rho, "data") else l.rho)[:] = st[2]
    if cor:
        g.set_coriolis(*cor); o.set_coriolis(*cor)
    steps = int(rng.integers(1, 6))
    g.run(steps); o.run(steps)
    g.u.read_from_device(); g.rho.read_from_device()
    fi = np.asarray(g.download_fi())
    ref = np.asarray(o.fi)
    if fi.dtype == np.uint16:            # value equality: the FP16C codes 0x0000 and 0x8000 are both zero
        fi = fi.copy(); ref = ref.copy(); fi[fi == 0x8000] = 0; ref[ref == 0x8000] = 0
    ok = np.array_equal(g.u.data, o.u) and np.array_equal(g.rho.data, o.rho) and np.array_equal(fi.ravel(), ref.ravel())
    print("%3d  %4dx%2dx%2d %s kernel %d shell %-4s force %d coriolis %d steps %d : %s" % (k, Nx, Ny, Nz, "fp16c" if fp16c else "f32  ", kern, shell,
        any(force), bool(cor), steps, "ok" if ok else "MISMATCH"), flush=True)
    g.close()
    assert ok
print("all %d cases identical" % cases)
