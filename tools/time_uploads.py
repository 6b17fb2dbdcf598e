#!/usr/bin/env python3
"""host <-> device copy rates of the lattice arrays (GPU box): single 512^3 solver and the domains of a group"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
L = luw.load()
def t(f, label, nbytes):
    t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
    print("%-56s %8.3f s  %7.2f GB/s" % (label, dt, nbytes / dt / 1e9), flush=True)
g = luw.LBM(512, 512, 512, 1e-4)
N = 512 ** 3
for rep in range(2):
    t(lambda: capi.check(L.luw_upload(g._h, capi.MASK_U)), "single 512^3: upload u (rep %d)" % rep, 12 * N)
    t(lambda: capi.check(L.luw_upload(g._h, capi.MASK_RHO)), "single 512^3: upload rho", 4 * N)
    t(lambda: capi.check(L.luw_upload(g._h, capi.MASK_FLAGS)), "single 512^3: upload flags", N)
    t(lambda: capi.check(L.luw_download(g._h, capi.MASK_U)), "single 512^3: download u", 12 * N)
    t(lambda: capi.check(L.luw_download(g._h, capi.MASK_FLAGS)), "single 512^3: download flags", N)
g.close()
gr = luw.LBMGroup(1024, 512, 512, 2, 1, 1, 1e-4, devices=[0, 0])
for d in range(2):
    h = L.luw_group_domain(gr._h, d)
    lN = gr.domain_info(d)[0]; n = lN[0] * lN[1] * lN[2]
    t(lambda: capi.check(L.luw_upload(h, capi.MASK_U)), "group [2,1,1] domain %d (%dx%dx%d): upload u" % (d, *lN), 12 * n)
    t(lambda: capi.check(L.luw_upload(h, capi.MASK_FLAGS)), "  upload flags", n)
    t(lambda: capi.check(L.luw_upload(h, capi.MASK_RHO)), "  upload rho", 4 * n)
    t(lambda: capi.check(L.luw_download(h, capi.MASK_U)), "  download u", 12 * n)
gr.close()
