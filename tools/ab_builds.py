#!/usr/bin/env python3
"""Interleaved A/B of several BUILDS of libluw_core.so in ONE process on one GPU: every build is compiled here (hipcc, the product flags plus the
build's extra flags) into gpurun_out/ab/, loaded side by side through the bare C-ABI, and the workloads are stepped round-robin -- build A, build B,
build A ... -- so that the drift of a box (whole sessions run 10 % apart, profiles/r02_skew_study.md) hits all builds alike.
usage: ab_builds.py "<tag>=<extra flags>" ... [-- workload ...]      workload = f32|fp16c : NxxNyxNz [: bld] [: cor] [: urban] [: th] [: nat]
       (urban = the 8-GPU tile's forcing: buffer nudging 80 cells + top sponge 100 layers; bld = building array; cor = Coriolis; th = thermal lattice)
e.g.   ab_builds.py "lib=" "noplain=-DLUW_PLAIN_ARITH=0" -- fp16c:1024x1024x256:bld fp16c:1024x1024x256:bld:cor fp16c:512x512x512:bld:urban:cor"""
import ctypes as C, os, subprocess, sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime for everything, see capi.load)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latticeurbanwind_amd.capi import Config
from bench import fill_channel, tile_forcing, coriolis_omega, NU

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
builds = [a.split("=", 1) for a in args[:cut]]
workloads = args[cut + 1:] or ["fp16c:1024x1024x256:bld", "fp16c:1024x1024x256:bld:cor", "fp16c:512x512x512"]
ROUNDS, STEPS = int(os.environ.get("ROUNDS", "5")), int(os.environ.get("STEPS", "40"))
out_dir = os.path.join(ROOT, "gpurun_out", "ab"); os.makedirs(out_dir, exist_ok=True)
src = os.path.join(ROOT, "latticeurbanwind_amd", "csrc", "luw_core.hip")
libs = []
for tag, extra in builds:
    so = os.path.join(out_dir, "libluw_%s.so" % tag)
    if extra.startswith("@"):                     # a library built elsewhere (e.g. an older commit, shipped under tools/): "<tag>=@<path>"
        so = os.path.join(ROOT, extra[1:])
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17", "-Wno-unused-function", *extra.split(),
        "-shared", "-o", so, src]
    if not extra.startswith("@"):
        subprocess.check_call(cmd)
    L = C.CDLL(so)
    L.luw_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]; L.luw_host_ptr.argtypes = [C.c_void_p, C.c_int]; L.luw_host_ptr.restype = C.c_void_p
    L.luw_run_timed.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]; L.luw_initialize.argtypes = [C.c_void_p]; L.luw_run.argtypes = [C.c_void_p,
        C.c_uint64]
    L.luw_set_coriolis.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]; L.luw_destroy.argtypes = [C.c_void_p]; L.luw_last_error.restype = C.c_char_p
    libs.append((tag, L))


def make(L, fp16c, N, opts):
    cfg = Config(); cfg.struct_size = C.sizeof(Config); cfg.Nx, cfg.Ny, cfg.Nz = N; cfg.Dx = cfg.Dy = cfg.Dz = 1; cfg.nu = NU
    cfg.ddf_format = 1 if fp16c else 0
    if "th" in opts: cfg.options = 8; cfg.alpha = 2.1e-7
    if "nat" in opts: cfg.options |= 16          # LUW_OPT_NATIVE_ARITH
    if "urban" in opts:
        nud, spg = tile_forcing()
        cfg.buffer_nudging_active = 1; cfg.buffer_n_cells = nud["n_cells"]; cfg.buffer_inv_tau_lbmu = nud["inv_tau"]; cfg.buffer_downstream_face_id = nud[
            "downstream_face"]
        cfg.top_sponge_active = 1; cfg.sponge_n_cells = spg["n_cells"]; cfg.sponge_inv_tau_lbmu = spg["inv_tau"]
    h = C.c_void_p()
    assert L.luw_create(C.byref(cfg), C.byref(h)) == 0, L.luw_last_error()
    n = N[0] * N[1] * N[2]
    view = lambda field, ct, count: np.ctypeslib.as_array(C.cast(L.luw_host_ptr(h, field), C.POINTER(ct)), (count,))
    fill_channel(view(2, C.c_uint8, n), view(1, C.c_float, 3 * n), view(0, C.c_float, n), *N, buildings="bld" in opts)
    if "cor" in opts: assert L.luw_set_coriolis(h, *coriolis_omega()) == 0
    assert L.luw_initialize(h) == 0; assert L.luw_run(h, 5) == 0
    return h


for w in workloads:
    parts = w.split(":"); fp16c = parts[0] == "fp16c"; N = tuple(int(v) for v in parts[1].split("x")); opts = set(parts[2:])
    hs = [make(L, fp16c, N, opts) for _, L in libs]
    res = [[] for _ in libs]
    for rnd in range(ROUNDS):
        for k, (_, L) in enumerate(libs):
            ms = C.c_double(); assert L.luw_run_timed(hs[k], STEPS, C.byref(ms)) == 0; res[k].append(ms.value)
    base = sorted(res[0])[len(res[0]) // 2]
    for (tag, L), r, h in zip(libs, res, hs):
        med = sorted(r)[len(r) // 2]
        print("%-44s %-14s kernel ms median %.4f min %.4f  (%+.1f %% vs %s)   rounds %s" % (w, tag, med, min(r), (med / base - 1) * 100, libs[0][0],
            " ".join("%.3f" % v for v in r)), flush=True)
        L.luw_destroy(h)
