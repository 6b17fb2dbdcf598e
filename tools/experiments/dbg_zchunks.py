import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import latticeurbanwind_amd as luw
luw.load()
import test_gpu_zchunks as T
own, D, fp16c = (130, 8, 12), (2, 1, 1), False
gN = tuple(o * d for o, d in zip(own, D))
ref, _ = T.run(luw, gN, D, fp16c, None, "sequential", (9,))
for seed in (0, 1, 2):
    got, used = T.run(luw, gN, D, fp16c, "zchunks", "batch", (9,), jitter=seed)
    from latticeurbanwind_amd.distributed import DomainLayout
    lay = DomainLayout(gN, D, 0); lx, ly, lz = lay.lN
    for name, a, b in zip(("u", "rho", "fi"), ref, got):
        d = np.nonzero(a != b)[0]
        if len(d):
            n = d % (lx * ly * lz)
            print("seed", seed, name, len(d), "cells x", sorted(set((n % lx).tolist()))[:12], "y", sorted(set(((n // lx) % ly).tolist())), "z", sorted(set((n // (lx * ly)).tolist())), "planes", sorted(set((d // (lx*ly*lz)).tolist()))[:19])
        else: print("seed", seed, name, "equal")
