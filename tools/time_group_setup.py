#!/usr/bin/env python3
"""Where the set-up time of a large multi-domain object goes (GPU box): create, scatter per field, initialise, download + gather.
usage: time_group_setup.py [gNx gNy gNz Dx Dy Dz]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
from bench import fill_channel, NU
L = luw.load()
a = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else [2048, 1024, 512, 4, 2, 1]
gN, D = tuple(a[:3]), tuple(a[3:])
n = D[0] * D[1] * D[2]
T = {}
def tick(name, t0): T[name] = time.perf_counter() - t0; print("%-44s %8.3f s" % (name, T[name]), flush=True)
t0 = time.perf_counter(); g = luw.LBMGroup(*gN, *D, NU, devices=[0] * n); tick("LBMGroup(): luw_group_create + numpy arrays", t0)
t0 = time.perf_counter(); fill_channel(g.flags, g.u, g.rho, *gN, buildings=True); tick("fill_channel (numpy, global arrays)", t0)
p = lambda x: x.ctypes.data_as(C.c_void_p)
for name, field, arr in (("flags", capi.FIELD_FLAGS, g.flags), ("u", capi.FIELD_U, g.u), ("rho", capi.FIELD_RHO, g.rho)):
    t0 = time.perf_counter(); capi.check(L.luw_group_scatter(g._h, field, p(arr))); tick("scatter " + name, t0)
t0 = time.perf_counter(); capi.check(L.luw_group_initialize(g._h)); tick("luw_group_initialize (upload + init kernel + exchange)", t0)
g._initialized = True
t0 = time.perf_counter(); g.run(4); tick("4 steps", t0)
t0 = time.perf_counter(); capi.check(L.luw_group_download(g._h, capi.MASK_U | capi.MASK_RHO)); tick("download u, rho", t0)
t0 = time.perf_counter(); capi.check(L.luw_group_gather(g._h, capi.FIELD_U, p(g.u))); capi.check(L.luw_group_gather(g._h, capi.FIELD_RHO, p(g.rho))); tick(
    "gather u, rho", t0)
t0 = time.perf_counter(); g.close(); tick("destroy", t0)
