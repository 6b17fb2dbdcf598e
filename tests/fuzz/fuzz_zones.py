#!/usr/bin/env python3
"""Random fuzz of the force zones against the CPU oracle (GPU box): buffer nudging and top sponge of random thickness (thin, thicker than a wave's
128 cells, W / E zones that overlap on narrow lattices), random downstream face and vertical nudging, odd and even rows, TYPE_E shell over a solid
ground, building solids, Coriolis and a volume force at random -- the general pair kernel with its early reference fetch, the FP16C and FP32
one-cell kernels.  Any mismatch in u, rho or a DDF plane stops the run.  usage: fuzz_zones.py [CASES] [SEED]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import latticeurbanwind_amd as luw
from oracle import oracle
from helpers import synthetic_state
from test_gpu_parity import make_pair, check


def run(cases, seed, say=print):
    rng = np.random.default_rng(seed)
    luw.load()
    for k in range(cases):
        Nx = int(rng.choice([rng.integers(128, 140), rng.integers(250, 300), rng.integers(500, 530), rng.integers(20, 60)]))
        Ny, Nz = int(rng.integers(8, 40)), int(rng.integers(8, 34))
        fp16c = bool(rng.integers(0, 4))
        kernel = "auto" if fp16c else "s"
        if fp16c and rng.integers(0, 5) == 0: kernel = "s"
        nb = int(rng.integers(1, max(2, min(Nx, Ny, Nz) // 2)))
        if rng.integers(0, 4) == 0: nb = int(rng.integers(1, 4))
        nud = dict(n_cells=nb, inv_tau=float(rng.uniform(0.002, 0.05)), downstream_face=int(rng.integers(0, 5)),
            nudge_vertical=int(rng.integers(0, 2))) if rng.integers(0, 6) else None
        spg = dict(n_cells=int(rng.integers(1, max(2, Nz - 3))), inv_tau=float(rng.uniform(0.002, 0.05))) if rng.integers(0, 5) else None
        cor = (0.0, 3e-5, 4e-5) if rng.integers(0, 2) else None
        force = tuple(float(v) for v in (rng.standard_normal(3) * 1e-5)) if rng.integers(0, 3) == 0 else (0.0, 0.0, 0.0)
        st = synthetic_state(Nx, Ny, Nz, seed=int(rng.integers(0, 1 << 30)), shell="luw", solids=bool(rng.integers(0, 2)))
        g, o = make_pair(luw, oracle, Nx, Ny, Nz, 2e-5, fp16c, kernel, st, force=force, coriolis=cor, nudging=nud, sponge=spg,
            every_step=bool(rng.integers(0, 2)))
        steps = int(rng.integers(1, 7))
        g.run(steps); o.run(steps)
        what = "case %d: %dx%dx%d %s kernel %s nudging %s sponge %s coriolis %s force %s steps %d" % (
            k, Nx, Ny, Nz, "fp16c" if fp16c else "f32", kernel, nud, spg, cor is not None, force != (0.0, 0.0, 0.0), steps)
        check(g, o, what)
        g.close()
        say("ok " + what)
    return cases


if __name__ == "__main__":
    n = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1, say=lambda t: print(t, flush=True))
    print("fuzz_zones: %d cases equal the oracle" % n)
