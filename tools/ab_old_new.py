#!/usr/bin/env python3
"""Interleaved A/B of two builds of libluw_core.so through the bare C-ABI (no package import): 512^3 and 1024x1024x256
FP32 channel, mean stream_collide time from luw_run_timed.  usage: ab_old_new.py libA.so libB.so"""
import ctypes as C, os, sys, time
import numpy as np
import torch  # noqa: F401  (one HIP runtime for everything, see capi.load)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticeurbanwind_amd.capi import Config
from bench import channel_state

def make(libpath, N):
    L = C.CDLL(libpath)
    L.luw_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]; L.luw_host_ptr.argtypes = [C.c_void_p, C.c_int]; L.luw_host_ptr.restype = C.c_void_p
    L.luw_run_timed.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]; L.luw_initialize.argtypes = [C.c_void_p]; L.luw_run.argtypes = [C.c_void_p,
        C.c_uint64]
    L.luw_last_error.restype = C.c_char_p
    cfg = Config(); cfg.struct_size = C.sizeof(Config); cfg.Nx, cfg.Ny, cfg.Nz = N; cfg.Dx = cfg.Dy = cfg.Dz = 1; cfg.nu = 1.48e-7
    cfg.ddf_format = 1 if FP16C else 0
    h = C.c_void_p()
    assert L.luw_create(C.byref(cfg), C.byref(h)) == 0, L.luw_last_error()
    n = N[0] * N[1] * N[2]
    fl, u, rho = channel_state(*N)
    for field, arr, ct in ((0, rho, C.c_float), (1, u, C.c_float), (2, fl, C.c_uint8)):
        p = L.luw_host_ptr(h, field); np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (arr.size,))[:] = arr
    assert L.luw_initialize(h) == 0; assert L.luw_run(h, 5) == 0
    return L, h

libs = sys.argv[1:3]
shapes = [(False, (512, 512, 512)), (True, (512, 512, 512)), (False, (1024, 1024, 256))]
if len(sys.argv) > 3:      # extra args: f32:NxxNyxNz or fp16c:NxxNyxNz
    shapes = [(a.split(":")[0] == "fp16c", tuple(int(v) for v in a.split(":")[1].split("x"))) for a in sys.argv[3:]]
for FP16C, N in shapes:
    objs = [make(p, N) for p in libs]
    res = [[] for _ in libs]
    for rnd in range(5):
        for k, (L, h) in enumerate(objs):
            ms = C.c_double(); assert L.luw_run_timed(h, 40, C.byref(ms)) == 0; res[k].append(ms.value)
    for p, r in zip(libs, res):
        r = sorted(r); print("%s %s %-28s kernel ms min %.3f median %.3f -> %.0f MLUPS" % (N, "fp16c" if FP16C else "f32", os.path.basename(p), r[0], r[2],
            N[0] * N[1] * N[2] / r[2] / 1e3))
    for L, h in objs:
        L.luw_destroy.argtypes = [C.c_void_p]; L.luw_destroy(h)
