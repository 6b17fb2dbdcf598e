#!/usr/bin/env python3
"""A/B of kernel variants on the SAME solver object (same device memory, so placement effects cancel): alternates
luw_set_kernel between the listed variants, 40 steps each, several rounds.
Needs the tools build of the library (make -C latticeurbanwind_amd/csrc ab -> tools/libluw_core_ab.so): the product library has no variants.
usage: ab_kernels.py [f32|fp16c] Nx Ny Nz kernelA kernelB ...   (names: scalar, general, cached, nt_all)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
from bench import channel_state
capi.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libluw_core_ab.so"))

NAMES = {"scalar": capi.KERNEL_SCALAR, "general": capi.KERNEL_SCALAR_GENERAL, "cached": capi.KERNEL_SCALAR_CACHED, "nt_all": capi.KERNEL_SCALAR_NT_ALL}
dt = sys.argv[1]; N = tuple(int(v) for v in sys.argv[2:5]); kernels = sys.argv[5:] or ["scalar", "general"]
g = luw.LBM(*N, 1.48e-7, fp16c=(dt == "fp16c"))
fl, u, rho = channel_state(*N, buildings="--buildings" in os.environ.get("AB_FLAGS", ""))
g.flags.data[:] = fl; g.u.data[:] = u; g.rho.data[:] = rho
g.run(0); g.run(10)
res = {k: [] for k in kernels}
for rnd in range(6):
    for k in kernels:
        capi.check(g._L.luw_set_kernel(g._h, NAMES[k]))
        res[k].append(g.run_timed(40))
for k in kernels:
    r = sorted(res[k]); print(
        "%s %s %-8s kernel ms min %.3f median %.3f -> %.0f MLUPS" % (N, dt, k, r[0], r[len(r) // 2], N[0] * N[1] * N[2] / r[len(r) // 2] / 1e3))
