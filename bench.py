#!/usr/bin/env python3
"""bench.py -- MLUPS of the D3Q19 collide-stream hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N = 1 (the driver's BENCH line): BASELINE.json configs[2] ("C3", the largest single-GPU configuration): 1024x1024x256 D3Q19,
voxelised building cluster (SURVEY 8d closed-form box array), bounce-back, Smagorinsky LES, FP32 DDFs, z=0 plane solid, the
other five outer faces TYPE_E with a log-law inflow profile, interior initialised with the same profile, rho=1.  The same JSON
line carries `secondary` blocks measured in the same process after the headline: configs[1] ("C2", 512^3 empty channel) and the
north star's 1024^3 grid, FP32 and FP16C each, and C3 with FP16C DDFs (--no-secondary skips them; --workload picks another
headline).
N > 1 (the driver's SCALE runs): the urban tile of BASELINE configs[3] weak-scaled at 512^3 cells per GPU (8 GPUs:
2048x1024x512) -- building array, buffer nudging and top sponge with the deck defaults (160 m / 300 s, 200 m / 120 s), cut
as n_gpu=[1,4,2] by default (--n-gpu 4 2 1 = the deck's literal grid, also measured as a secondary block), one-cell halos over
RCCL.  There is no fallback transport: if RCCL point-to-point fails the run exits non-zero.
A "step" is one stream_collide pass over the whole lattice; rho/u are written by the last step only (153 B/LUP mode, see
DESIGN.md); data is synthetic and resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); normally already exported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: gloo (side channel / test aid) over loopback, no hostname lookup

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_LUP = {"f32": 153.0, "fp16c": 77.0}   # 19 DDF reads + 19 DDF writes + 1 flag byte (FX/lbm.cpp:122)
NU = 1.48e-7                    # units.nu(1.48e-5) for cell = 2 m, U_ref = 10 m/s at u_lbm = 0.1
CELL_M, DT_S = 2.0, 2.0 * 0.1 / 10.0            # the unit system behind NU: 2 m cells, dt = cell * u_lbm / U_ref

WORKLOADS = {   # name -> (lattice, building array, BASELINE.json reference)
    "c3": ((1024, 1024, 256), True, "BASELINE configs[2]"),
    "c2": ((512, 512, 512), False, "BASELINE configs[1]"),
    "cube1024": ((1024, 1024, 1024), False, "north-star 1024^3-class grid"),
    "tile512": ((512, 512, 512), True,
        "one GPU's share of the BASELINE configs[3] / [4] urban tile (the N = 1 point of the N > 1 lines; --urban adds its nudging + sponge)"),
}


def loglaw_profile(nz, u_max=0.1):
    """lattice-unit inflow speed per z level: log law over the cell centres above the solid z=0 plane"""
    z = (np.arange(nz, dtype=np.float64) - 0.5) * 2.0          # metres above ground, cell = 2 m
    z0 = 0.3
    u = np.log(np.maximum(z, 0.0) / z0 + 1.0)
    u[0] = 0.0
    return (u_max * u / u.max()).astype(np.float32)


BUILDING_TOP = 1 + 8 + 47      # no building cell at or above this z (heights 8 + (.. mod 48), standing on the z=0 plane)


def fill_channel(flags, u, rho, Nx, Ny, Nz, gx0=0, gy0=0, gz0=0, GNx=None, GNy=None, GNz=None, buildings=False):
    """Writes flags/u/rho of the (sub)box [gx0,gx0+Nx) x ... of the global channel GNx x GNy x GNz (wind along +x) into the
    given arrays (reference layout, e.g. a solver's host mirrors), in place and without lattice-sized temporaries.
    buildings=True adds BASELINE configs[2]'s solid mask: an array of axis-aligned boxes, footprint 24x24 cells on a 64-cell
    pitch, heights 8+((7i+13j) mod 48) cells (closed form)"""
    GNx, GNy, GNz = GNx or Nx, GNy or Ny, GNz or Nz
    prof = loglaw_profile(GNz)
    zs = (np.arange(Nz) + gz0) % GNz; ys = (np.arange(Ny) + gy0) % GNy; xs = (np.arange(Nx) + gx0) % GNx
    fl3 = flags.reshape(Nz, Ny, Nx); u4 = u.reshape(3, Nz, Ny, Nx)
    fl3[:] = 0
    bz = (zs == GNz - 1); by = (ys == 0) | (ys == GNy - 1); bx = (xs == 0) | (xs == GNx - 1)
    fl3[bz, :, :] = 2; fl3[:, by, :] = 2; fl3[:, :, bx] = 2
    fl3[zs == 0, :, :] = 1
    low = np.nonzero(zs < BUILDING_TOP)[0]                                     # the only z levels that can hold solids
    if buildings and low.size:
        i, j = xs // 64, ys // 64
        inx = (xs % 64 >= 20) & (xs % 64 < 44) & (xs > 0) & (xs < GNx - 1); iny = (ys % 64 >= 20) & (ys % 64 < 44) & (ys > 0) & (ys < GNy - 1)
        h = 8 + ((7 * i[None, :] + 13 * j[:, None]) % 48)                       # (Ny, Nx) building height in cells
        zl = zs[low]
        solid = (inx[None, :] & iny[:, None])[None, :, :] & (zl[:, None, None] >= 1) & (zl[:, None, None] < 1 + h[None, :, :])
        sub = fl3[low]; sub[solid] = 1; fl3[low] = sub
    u4[0] = prof[zs][:, None, None]; u4[1] = 0.0; u4[2] = 0.0
    if low.size:
        sub = u4[0][low]; sub[fl3[low] == 1] = 0.0; u4[0][low] = sub
    rho[:] = 1.0


def channel_state(Nx, Ny, Nz, gx0=0, gy0=0, gz0=0, GNx=None, GNy=None, GNz=None, buildings=False):
    """fill_channel into fresh arrays: (flags, u, rho), flat, reference layout"""
    n = Nx * Ny * Nz
    flags, u, rho = np.empty(n, np.uint8), np.empty(3 * n, np.float32), np.empty(n, np.float32)
    fill_channel(flags, u, rho, Nx, Ny, Nz, gx0, gy0, gz0, GNx, GNy, GNz, buildings)
    return flags, u, rho


def tile_forcing():
    """buffer nudging + top sponge of the urban tile (BASELINE configs[3]/[4]) with the deck defaults of
    project_template/conf.luw:49-56 in the benchmark's unit system: Nbuf = round(160 m / cell), inv_tau = dt / 300 s
    (FX/setup.cpp:3844-3856), Nsponge = round(200 m / cell), inv_tau = dt / 120 s (:3867-3881); wind along +x, so the east face
    is the downstream one (:3756-3761)"""
    return (dict(n_cells=int(round(160.0 / CELL_M)), inv_tau=DT_S / 300.0, downstream_face=2, nudge_vertical=0),
            dict(n_cells=int(round(200.0 / CELL_M)), inv_tau=DT_S / 120.0))


def reference_case_rmse(case, ddf, fixture, arith="exact"):
    """the deck driver on one committed synthetic case (48x40x24 + sponge layers, 64 steps) against the fields the REAL reference wrote for the same deck
    on an MI355X (tests/golden/<fixture>.npz): u RMSE in lattice units over the non-solid cells at K = 8, K = 64 and of u_avg (mean of the last four steps)"""
    import glob, shutil, subprocess, tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from vtkio import read_vtk
    drv = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")
    gold = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
    tmp = tempfile.mkdtemp()
    try:
        shutil.copytree(os.path.join(ROOT, "tests", "golden", "refcases", case), os.path.join(tmp, case))
        r = subprocess.run([drv, os.path.join(tmp, case, "conf.luwpf"), "--ddf", ddf, "--arith", arith], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": "driver exit %d" % r.returncode}
        fac = np.float32(5.0) / np.float32(0.1)                      # si_ref_u = max profile U = 5 m/s, u_lbm = 0.1
        fluid = ~gold["solid"]

        def err(mine, ref):
            d = ((mine - ref) / fac)[fluid].astype(np.float64)
            return float(np.sqrt((d ** 2).sum(-1).mean()))
        vt = os.path.join(tmp, case, "RESULTS", "vtk")
        out = {}
        for t in (8, 64):
            out["K%d" % t] = err(read_vtk(glob.glob(os.path.join(vt, "*_raw_u-%09d.vtk" % t))[0])[1]["data"], gold["u%d" % t])
        out["u_avg"] = err(read_vtk(glob.glob(os.path.join(vt, "*_avg-000000064.vtk"))[0])[1]["u_avg"], gold["u_avg"])
        return {k: float("%.3e" % v) for k, v in out.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def reference_parity():
    """u-field RMSE against the REAL reference, measured now through the deck driver on this GPU: FP32 DDFs against the reference's FP32 build (case B: one
    building, LES), and the SHIPPED precision -- FP16C DDFs -- against the reference's shipped build (case A: LES with nudging + sponge; case L: laminar),
    with the bit-exact kernels and with the native-arithmetic ones (--arith native)."""
    fp32 = reference_case_rmse("CaseB", "fp32", "ref_fp32_CaseB")
    shipped = {}
    for case in ("CaseA", "CaseL"):
        shipped[case] = {"exact": reference_case_rmse(case, "fp16c", "ref_shipped_" + case), "native": reference_case_rmse(case, "fp16c", "ref_shipped_" + case,
            "native")}
    k64 = fp32.get("K64")
    return {"u_rmse_vs_reference": k64, "unit": "lattice units", "steps": 64, "tolerance": 1e-5,
            "case": "tests/golden/refcases/CaseB (48x40x24, FP32 DDFs)", "lattice": [48, 40, 24], "cells": 48 * 40 * 24,
            "fp32": dict(fp32, case="CaseB", within_tolerance=bool(k64 is not None and k64 < 1e-5)),
            "shipped": dict(shipped, precision="FP16C DDFs (what the reference ships), reference build FP16C + TEMPERATURE",
                within_tolerance_at_K8=all(v[a].get("K8", 1.0) < 1e-5 for v in shipped.values() for a in v),
                within_tolerance_at_K64=all(v[a].get("K64", 1.0) < 1e-5 for v in shipped.values() for a in v),
                note="FP16C storage rounds every stored population to 2^-12 relative; any arithmetic that is not bit-identical to the reference's own (built "
                    "by "
                     "the OpenCL driver with -cl-mad-enable and native division, i.e. not bit-defined) flips single roundings, which LES flow then amplifies: "
                         "at "
                     "K = 64 the LES case A sits at 2.6e-5 (u_avg 1.5e-5) -- OUTSIDE the north star's 1e-5 -- for the bit-exact kernels, the CPU oracle and "
                     "the native-arithmetic kernels alike; the laminar case L (4e-6) and every case at K = 8 (< 1e-6) are inside.  FP32 DDFs: 1.2e-7."),
            "horizon": "K = 64 steps on 46 k cells is the ONLY horizon pinned by outputs of the real reference (17 committed cases, FP32 and shipped FP16C "
                "builds, tests/golden/ref_*.npz); beyond it the chain is HIP path == CPU oracle bit for bit (literal 128^3 configs[0] at K = 100 turbulent "
                "and K = 1000 laminar, tests/test_gpu_c1.py; the bench workloads at full size, tests/test_gpu_bench_workloads.py) and oracle vs reference "
                "0.5-1.3e-7 (FP32) at K = 64"}


def coriolis_omega():
    """Omega_earth (0, cos phi, sin phi) dt in lattice units at 31.25 deg N for cell = 2 m, U_ref = 10 m/s, u_lbm = 0.1 (FX/setup.cpp:3800-3823)"""
    import math
    return 0.0, 7.292115e-5 * math.cos(math.radians(31.25)) * DT_S, 7.292115e-5 * math.sin(math.radians(31.25)) * DT_S


def usable_cores():
    """cores this process may really use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    whole host inside containers)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(max_seconds=20.0):
    """The CPU oracle (our C/OpenMP restatement of the reference kernel, kind "port"; its row-wise path: the same operations per cell as the literal
    path, eight cells per AVX2 statement, indices by addition -- oracle/luw_oracle.c, bit-identical to the literal path) timed on this box's host
    cores on a bounded sample of the benchmark recipe: 256^3 channel, FP32 DDFs, as many steps as fit into max_seconds.  The thread
    count is the fastest of a short sweep (over-subscription inside a CPU-limited container is disastrous).  Reported with the
    CPU model and the DRAM bandwidth it amounts to: the restatement moves 169 B per lattice update like the reference's
    UPDATE_FIELDS kernel (153 + 16 for rho,u every step), set against a copy-kernel bandwidth measured with the same threads."""
    from oracle import oracle
    N = 256
    o = oracle.OracleLBM(N, N, N, NU)
    fill_channel(o.flags, o.u, o.rho, N, N, N)
    o.run(1)
    cores = usable_cores()
    cands = sorted({max(1, c) for c in (cores, cores // 2, cores // 4, 64, 32, 16, 8) if c <= cores}, reverse=True)
    best_t, best_rate = cands[-1], 0.0
    for t in cands:
        oracle.set_threads(t)
        t0 = time.perf_counter(); o.run(2); dt = time.perf_counter() - t0
        if 2 * N ** 3 / dt > best_rate:
            best_rate, best_t = 2 * N ** 3 / dt, t
    oracle.set_threads(best_t)
    steps, t0 = 0, time.perf_counter()
    while True:
        o.run(2); steps += 2
        dt = time.perf_counter() - t0
        if dt > max_seconds or steps >= 2000:
            break
    mlups = N ** 3 * steps / dt / 1e6
    copy_gbps = oracle.copy_bandwidth_gbps(1 << 30)          # read 1 GiB + write 1 GiB with the same OpenMP threads
    return {"value": round(mlups, 1), "unit": "MLUPS", "cores": best_t, "kind": "port", "cpu_model": cpu_model(),
            "dram_GBps": round(mlups * 169.0 / 1e3, 1), "copy_bandwidth_GBps": round(copy_gbps, 1),
                "dram_frac_of_copy": round(mlups * 169.0 / 1e3 / copy_gbps, 3) if copy_gbps else None,
            "path": "row-wise (AVX2, 8 cells per statement; bit-identical to the literal one-cell-at-a-time path)" if oracle.fast_available() else "literal",
            "sample": "%d steps of a 256^3 FP32 channel (same recipe as the GPU workloads, 169 B per update incl. rho,u every step) in %.1f s, OpenMP threads "
                "swept over %s of %d usable cores" % (steps, dt, cands, cores)}


def device_context(torch, device):
    """What this particular GPU streams by itself, next to the contract's 8 TB/s: the same binary ran the HBM-bound FP32 step 12 % apart on
    different MI355X boxes of the pool (profiles/r02_skew_study.md), so the line carries the box's own device-to-device copy rate (2 GiB
    tensor copy, bytes read + written, best of 5) and the memory / fabric clock levels the driver reports right after it."""
    ctx = {"name": torch.cuda.get_device_name(device)}
    try:
        n = 1 << 29                                               # 2 GiB of float32
        src = torch.empty(n, dtype=torch.float32, device="cuda:%d" % device).fill_(1.0)
        dst = torch.empty_like(src)
        best = None
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); dst.copy_(src); e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        ctx["copy_GBps"] = round(2.0 * n * 4 / (best * 1e-3) / 1e9, 1)
        ctx["copy_frac_of_peak"] = round(ctx["copy_GBps"] / HBM_PEAK_GBPS, 4)
        del src, dst
        torch.cuda.empty_cache()
    except Exception as e:
        ctx["copy_error"] = str(e)[:120]
    import glob
    for name in ("mclk", "fclk", "sclk"):
        for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_%s" % name)):
            try:
                cur = [l.strip() for l in open(path) if l.strip().endswith("*")]
                if cur:
                    ctx.setdefault(name, cur[0].rstrip("*").strip())
            except OSError:
                pass
    return ctx


def profile_key(dtype, size, buildings, coriolis=False, thermal=False, every_step=False, urban=False, native=False):
    return "%s_%dx%dx%d%s%s%s%s%s%s" % (dtype, size[0], size[1], size[2], "_bld" if buildings else "", "_urban" if urban else "", "_cor" if coriolis else "",
        "_th" if thermal else "", "_uf" if every_step else "", "_nat" if native else "")


def attach_traffic(roof, key, kernel):
    """HBM traffic of the dominant kernel from rocprofv3 PMC counters: collected in separate --pmc passes of this same workload
    (tools/profile_bench.sh), corrected as MI355X_MICROARCH.md prescribes (read requests are 128 B), committed under profiles/
    and keyed on the full configuration; null when no profile of exactly this workload exists"""
    # newest round first
    prof = next((q for q in (os.path.join(ROOT, "profiles", "r%02d_%s_summary.json" % (rnd, key)) for rnd in (4, 3, 2)) if os.path.exists(q)), None)
    if kernel == "auto" and prof:
        pr = json.load(open(prof))
        roof["traffic"] = round(pr["hbm_traffic_bytes_per_launch"])
        roof["traffic_source"] = "profiles/" + os.path.basename(prof) + " (TCC_EA0_RDREQ x 128 B + WRITE_SIZE x 1024, per launch)"


def run_single(luw, kern, device, size, dtype, buildings, steps, warmup, coriolis=False, thermal=False, every_step=False, kernel_name="auto", keep=None,
        urban=False, native=False):
    """one single-GPU workload: create, fill the host mirrors in place, upload + initialise, W warm-up steps, K timed steps.
    Returns the measurement block (MLUPS, ms/step, roofline of the stream_collide kernel)."""
    import torch
    Nx, Ny, Nz = size
    fp16c = dtype == "fp16c"
    # urban: buffer nudging + top sponge of the 8-GPU tile (general kernel on two thirds of the cells)
    nud, spg = tile_forcing() if urban else (None, None)
    lbm = luw.LBM(Nx, Ny, Nz, NU, fp16c=fp16c, kernel=kern, device=device, update_fields_every_step=every_step, alpha=(2.1e-7 if thermal else None),
        buffer_nudging=nud, top_sponge=spg, native_arith=native)
    try:
        fill_channel(lbm.flags.data, lbm.u.data, lbm.rho.data, Nx, Ny, Nz, buildings=buildings)
        solid = int(np.count_nonzero((lbm.flags.data & 3) == 1))
        if coriolis:
            lbm.set_coriolis(*coriolis_omega())
        from latticeurbanwind_amd import capi as _capi
        placement = _capi.placement_info(lbm._h)     # what luw_create's placement search did
        lbm.run(0)
        lbm.run(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms = lbm.run_timed(steps)             # K steps, HIP events around each launch on the launch stream
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        lbm.close()
    cells = Nx * Ny * Nz
    # + gi read / write; T, like rho and u, is stored by the last step of a run only unless every step is asked for (+ 4 B then)
    bpl = BYTES_PER_LUP[dtype] + (16.0 if every_step else 0.0) + ((7 * 2 * (2 if fp16c else 4) + (4 if every_step else 0)) if thermal else 0.0)
    # algorithmic bytes of one launch: every cell's flag byte is read; only non-solid cells (fluid and TYPE_E) move their DDFs
    launch_bytes = (bpl - 1.0) * (cells - solid) + 1.0 * cells
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
            "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": int(launch_bytes),
            "note": "achieved = (%g B x %d non-solid cells + 1 flag byte x %d solid cells) / mean stream_collide duration (HIP events on the launch stream)" % (
                bpl, cells - solid, solid)}
    attach_traffic(roof, profile_key(dtype, size, buildings, coriolis, thermal, every_step, urban, native and fp16c), kernel_name)
    mlups = cells * steps / dt / 1e6
    return {"value": round(mlups, 1), "unit": "MLUPS", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup,
            "lattice": [Nx, Ny, Nz], "dtype": "f32" if not fp16c else "fp16c-storage/f32-arithmetic", "solid_fraction": round(solid / cells, 5),
                "bytes_per_lup": bpl, "arith": "native" if (native and fp16c) else "exact", "create_s": placement["create_s"], "placement": placement,
            "options": ("building array" if buildings else "no solids above the ground plane")
                + (" + buffer nudging (160 m / 300 s) + top sponge (200 m / 120 s)" if urban else "") + (" + Coriolis force" if coriolis else "")
                + (" + thermal D3Q7 lattice (T stored with rho/u)" if thermal else "") + (", rho/u written every step" if every_step else ""),
            "roofline": roof}


SECONDARY_STEPS, SECONDARY_WARMUP = 200, 20
# the single-GPU secondary blocks: (workload, dtype, Coriolis, thermal lattice, urban forcing)
SINGLE_BLOCKS = {"c2_f32": ("c2", "f32", False, False, False), "c2_fp16c": ("c2", "fp16c", False, False, False),
    "c3_fp16c": ("c3", "fp16c", False, False, False),
                 "c3_fp16c_coriolis": ("c3", "fp16c", True, False, False),
                 "c3_fp16c_thermal": ("c3", "fp16c", False, True, False),      # FP16C DDFs + the thermal D3Q7 lattice: what the shipped reference build runs
                 "cube1024_f32": ("cube1024", "f32", False, False, False), "cube1024_fp16c": ("cube1024", "fp16c", False, False, False),
                 # the 8-GPU tile of BASELINE configs[3] / configs[4] seen from ONE GPU: its N = 1 point (512^3 urban tile, undivided)
                 "tile512_urban_f32": ("tile512", "f32", False, False, True), "tile512_urban_fp16c_coriolis": ("tile512", "fp16c", True, False, True)}


# FP16C blocks that are also measured with the native-arithmetic kernels (the forced / thermal / zone kernels the exact arithmetic costs most, and the
# plain one as the reference point)
NATIVE_TWINS = ("c3_fp16c", "c3_fp16c_coriolis", "c3_fp16c_thermal", "tile512_urban_fp16c_coriolis", "c5_rank_4x2x1_fp16c_coriolis")


def run_single_block(luw, capi, device, key, native=False):
    wl, dt_, cor, th, urban = SINGLE_BLOCKS[key]
    sz, bld, _ = WORKLOADS[wl]
    # >= 200 timed after >= 20 warm-up steps
    r = run_single(luw, capi.KERNEL_AUTO, device, sz, dt_, bld, SECONDARY_STEPS, SECONDARY_WARMUP, coriolis=cor, thermal=th, urban=urban, native=native)
    r["workload"] = (
        "512^3 urban tile, undivided: the N = 1 point of the N > 1 lines (BASELINE configs[3]%s per GPU)" % (" / configs[4]" if cor else "")) if urban \
        else describe(wl, sz, bld, dt_, cor, th, False)
    return r


RANK_SHAPE_BLOCKS = {"c4_rank_4x2x1_f32": dict(fp16c=False, coriolis=False, D=(4, 2, 1), rank=0),
                     "c5_rank_4x2x1_fp16c_coriolis": dict(fp16c=True, coriolis=True, D=(4, 2, 1), rank=0),
                     "c5_rank_1x4x2_fp16c_coriolis": dict(fp16c=True, coriolis=True, D=(1, 4, 2), rank=7)}


def run_rank_shape(luw, torch, kern, device, fp16c, coriolis, D, rank, steps, warmup, native=False, transport="rccl-self"):
    """One rank of the 2048x1024x512 urban tile (BASELINE configs[3]; configs[4] with FP16C + Coriolis) cut as n_gpu = D, in its real local shape and
    with its real share of the nudging / sponge zones, stepped through the production schedule of a multi-GPU run -- boundary shell on the
    communication stream, pack / exchange / unpack, interior on the compute stream, pipelined steps -- with every face going through the real
    transport to the rank itself (RCCL self send / receive).  D = (1,1,1): the 512^3 tile undivided, the N = 1 point of the scaling curve."""
    import torch.distributed as dist
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, DomainLayout, SelfExchangeTransport, PeerLoopbackTransport, init_rccl_process_group
    if transport == "rccl-self" and not dist.is_initialized():
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29539"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", str(device))):
            os.environ.setdefault(k, v)
        init_rccl_process_group(device)
    world = D[0] * D[1] * D[2]
    gN = (512 * D[0], 512 * D[1], 512 * D[2]) if world > 1 else (512, 512, 512)
    nud, spg = tile_forcing()
    lay = DomainLayout(gN, D, rank)
    tr = SelfExchangeTransport(lay) if transport == "rccl-self" else PeerLoopbackTransport(lay)
    if world > 1:
        tr.warm_up(torch.device("cuda", device), torch.float16 if fp16c else torch.float32)     # connections first, lattice second (as in a real run)
    sim = DomainDecomposedLBM(gN, D, NU, rank=rank, transport=tr, fp16c=fp16c, kernel=kern, device=device, buffer_nudging=nud, top_sponge=spg,
        native_arith=native)
    try:
        lb = sim.backend.lbm
        fill_channel(lb.flags.data, lb.u.data, lb.rho.data, *sim.layout.lN, *sim.layout.O, *gN, buildings=True)
        if coriolis:
            sim.backend.set_coriolis(*coriolis_omega())
        sim.initialize(); sim.run(warmup)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tm = sim.run(steps, timed=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        b = sim.layout.interior_box() if sim.overlap else sim.layout.whole_box()
        kcells = (b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4])
    finally:
        sim.backend.close()
    owned = (gN[0] // D[0]) * (gN[1] // D[1]) * (gN[2] // D[2])
    bpl = BYTES_PER_LUP["fp16c" if fp16c else "f32"]
    ms = dt / steps * 1e3
    return {"value": round(owned / (ms * 1e-3) / 1e6, 1), "unit": "MLUPS (this rank's owned cells per wall second)", "ms_per_step": round(ms, 4),
        "steps": steps, "warmup": warmup,
            "n_gpu": list(D), "rank": rank, "coord": list(sim.layout.coord), "local_lattice": list(sim.layout.lN), "dtype": "fp16c-storage/f32-arithmetic"
                if fp16c else "f32", "arith": "native" if (native and fp16c) else "exact",
            "workload": "rank %d of the 2048x1024x512 urban tile as n_gpu=%s" % (rank, list(D)) if world > 1
                else "512^3 urban tile, undivided (the N = 1 point of the N > 1 lines)",
            "options": "building array + buffer nudging (160 m / 300 s) + top sponge (200 m / 120 s)" + (" + Coriolis force" if coriolis else ""),
            "halo_exchange": None if world == 1 else "RCCL self send / receive of every face (no wire to another device)" if transport == "rccl-self"
                else "peer-loopback: the faces are written straight into the receive buffers the unpack reads (the one-process host's peer stores with the "
                     "rank as its own neighbour; no copy kernel, no wire)", "transport": transport, "overlap": bool(sim.overlap),
            "exchange": ("one batch per step: faces of all axes + the 12 edge populations" if sim.one_phase
                else "three phases x, y, z with rims (LUW_EXCHANGE)")
                if world > 1 else None, "x_faces": None if sim.layout.D[0] == 1 else "written by the step kernels" + (", read from the receive buffers by the "
                "next step's" if getattr(sim.backend, "x_insert_fused", False) else "; unpack kernel"),
            "kernel_ms": round(tm["kernel_ms"], 4) if tm else None, "shell_ms": None if not tm or tm.get("shell_ms") is None else round(tm["shell_ms"], 4),
            "exchange_ms": None if not tm or tm.get("exchange_ms") is None else round(tm["exchange_ms"], 4),
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "achieved": round(bpl * owned / (ms * 1e-3) / 1e9, 1),
                "frac": round(bpl * owned / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "kernel_frac": round(bpl * kcells / (tm["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if tm and tm.get("kernel_ms") else None,
                             "traffic": None,
                         "note": "frac = %g B/LUP x %d owned cells / wall time of a whole step (shell + exchange + interior); kernel_frac = the %s kernel "
                             "alone over its %d cells" % (bpl, owned, "interior-box" if sim.overlap else "whole-box", kcells)}}


def describe(name, size, buildings, dtype, coriolis, thermal, every_step):
    Nx, Ny, Nz = size
    what = "voxelised building cluster (closed-form box array, SURVEY 8d), bounce-back" if buildings else "empty channel"
    ref = WORKLOADS[name][2] if name in WORKLOADS and tuple(WORKLOADS[name][0]) == tuple(size) and WORKLOADS[name][1] == buildings else "custom size"
    return "%dx%dx%d D3Q19 %s (%s), log-law profile inflow on TYPE_E faces, solid ground, SRT+Smagorinsky LES, %s DDFs%s%s, rho/u written %s" % (
        Nx, Ny, Nz, what, ref, "FP16C" if dtype == "fp16c" else "FP32", " + Coriolis force" if coriolis else "", " + thermal D3Q7 lattice" if thermal else "",
        "every step" if every_step else "by the last step only")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3",
        help="N = 1 headline workload (default c3 = BASELINE configs[2], the largest single-GPU configuration)")
    ap.add_argument("--size", type=int, nargs=3, default=None,
        help="N = 1: lattice instead of the workload's; N > 1: per-GPU block (default 512 512 512, 8 GPUs = the 2048x1024x512 tile)")
    ap.add_argument("--n-gpu", type=int, nargs=3, default=None,
        help="domain grid Dx Dy Dz (default: x kept whole, e.g. 1 4 2 on 8 GPUs; the deck's literal 4 2 1 is accepted)")
    ap.add_argument("--dtype", choices=["f32", "fp16c"], default="f32")
    ap.add_argument("--kernel", choices=["auto", "scalar", "pair"], default="auto")
    ap.add_argument("--arith", choices=["exact", "native"], default="exact",
        help="FP16C collision arithmetic: exact = the bit-exact kernels (default, equal to the CPU oracle), native = LUW_OPT_NATIVE_ARITH (v_rcp / v_sqrt, "
             "free contraction; within the tolerance gates of tests/test_gpu_native_arith.py)")
    ap.add_argument("--buildings", action="store_true",
        help="add the configs[2] solid mask to a workload that has none (c3 and the N > 1 tile always carry it)")
    ap.add_argument("--no-buildings", action="store_true", help="N > 1: plain channel tile without the building array / nudging / sponge")
    ap.add_argument("--coriolis", action="store_true", help="Coriolis body force at 31.25 deg N (BASELINE configs[4]): every cell takes the forced path")
    ap.add_argument("--thermal", action="store_true", help="also run the thermal D3Q7 lattice (the shipped reference build always does): +7 DDF planes and T")
    ap.add_argument("--urban", action="store_true", help="N = 1: add the urban tile's buffer nudging + top sponge (deck defaults) to the workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the headline measurement (profiling runs)")
    ap.add_argument("--share-device", type=int, default=None, help="test aid: all ranks use this one GPU, halos through gloo + host staging (plumbing check of "
        "the N > 1 path on a 1-GPU box; the line is labelled, never a multi-GPU result)")
    ap.add_argument("--force-distributed", action="store_true",
        help="take the N > 1 code path (process group, DomainDecomposedLBM) even with one rank: plumbing check")
    ap.add_argument("--every-step-fields", action="store_true", help="write rho,u every step like the reference's UPDATE_FIELDS (169 B/LUP)")
    ap.add_argument("--secondary-block", choices=sorted(SINGLE_BLOCKS), default=None,
        help="(used by the N = 1 line itself) measure ONE single-GPU secondary block in this fresh process and print it")
    ap.add_argument("--rank-shape-block", choices=sorted(RANK_SHAPE_BLOCKS), default=None,
        help="(used by the N = 1 line itself) measure ONE rank-shape secondary block in this fresh process and print it")
    ap.add_argument("--rank-transport", choices=["rccl-self", "peer-loopback"], default="rccl-self",
        help="--rank-shape-block: how the rank's faces come back to it (RCCL self send / receive, or written in place like the one-process host's peer stores)")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip the self-check against the CPU oracle (profiling runs)")
    ap.add_argument("--no-group-host", action="store_true", help="N > 1: skip the one-process multi-domain host block")
    ap.add_argument("--group-host-child", choices=sorted(GROUP_HOST_VARIANTS), default=None,
        help="(used by the N > 1 line itself) measure ONE variant of the one-process host in this fresh process and print it")
    ap.add_argument("--devices", default=None, help="--group-host-child: the devices of the domains, comma separated")
    ap.add_argument("--global-lattice", type=int, nargs=3, default=None, help="--group-host-child: the whole lattice")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON, rank 0): everything libraries print on file descriptor 1 while the run is set up
    # (RCCL's "Librccl path", Gloo's connection notes) is sent to stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import latticeurbanwind_amd as luw
    from latticeurbanwind_amd import capi
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the hot path has no CPU fallback")
    if args.share_device is not None:
        local_rank = args.share_device
        os.environ.setdefault("LUW_TUNE_PLACEMENT", "0")    # rank processes on one device: their placement probes would time each other (luw_placement.hpp)
    torch.cuda.set_device(local_rank)
    luw.load()
    kern = {"auto": capi.KERNEL_AUTO, "scalar": capi.KERNEL_SCALAR, "pair": capi.KERNEL_PAIR}[args.kernel]
    fp16c = args.dtype == "fp16c"
    METRIC = "MLUPS (D3Q19) at 1/2/4/8 MI355X; % of HBM roofline; u-field RMSE vs ref"

    if args.group_host_child:
        try:
            blk = group_host_child(args, luw, capi, kern, fp16c)
        except Exception as e:
            blk = {"error": str(e)[:300]}
        sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(blk)); sys.stdout.flush()
        os.dup2(2, 1)
        return
    if args.secondary_block:
        try:
            blk = run_single_block(luw, capi, local_rank, args.secondary_block, native=args.arith == "native")
        except Exception as e:
            blk = {"error": str(e)[:300]}
        sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(blk)); sys.stdout.flush()
        os.dup2(2, 1)
        return
    if args.rank_shape_block:
        blk = run_rank_shape(luw, torch, capi.KERNEL_AUTO, local_rank, steps=args.steps, warmup=args.warmup, native=args.arith == "native",
            transport=os.environ.get("BENCH_RANK_TRANSPORT", args.rank_transport), **RANK_SHAPE_BLOCKS[args.rank_shape_block])
        import torch.distributed as dist
        if dist.is_initialized(): dist.destroy_process_group()
        sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(blk)); sys.stdout.flush()
        os.dup2(2, 1)
        return
    if world == 1 and not args.force_distributed:
        size, buildings, _ = WORKLOADS[args.workload]
        if args.size: size = tuple(args.size)
        buildings = buildings or args.buildings
        head = run_single(luw, kern, local_rank, size, args.dtype, buildings, args.steps, args.warmup, args.coriolis, args.thermal, args.every_step_fields,
            args.kernel, urban=args.urban, native=args.arith == "native")
        out = {
            "metric": METRIC, "value": head["value"], "unit": "MLUPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": describe(args.workload, size, buildings, args.dtype, args.coriolis, args.thermal, args.every_step_fields)
                + (", buffer nudging (160 m / 300 s) + top sponge (200 m / 120 s)" if args.urban else ""),
                       "global_lattice": list(size), "n_gpu": [1, 1, 1], "halo_exchange": None, "kernel": args.kernel, "bytes_per_lup": head["bytes_per_lup"],
                           "solid_fraction": head["solid_fraction"], "arith": head["arith"], "create_s": head["create_s"], "placement": head["placement"]},
            "roofline": dict(head["roofline"],
                whole_job_frac=round(head["roofline"]["algorithmic_bytes_per_launch"] / (head["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)),
        }
        if args.steps < 200:
            out["timed_region_note"] = ("%d timed steps (%.2f s) after %d warm-up steps: SURVEY 8(d) asks for >= 200 timed after >= 20 warm-up steps, which is "
                "what the secondary blocks run (%d / %d); the rate of this HBM-bound step does not depend on the length of the region beyond ~20 steps "
                "(profiles/r04_steps_sweep.txt)" % (args.steps, head["ms_per_step"] * args.steps * 1e-3, args.warmup, SECONDARY_STEPS, SECONDARY_WARMUP))
        out["device"] = device_context(torch, local_rank)
        if out["device"].get("copy_GBps"):
            out["roofline"]["frac_of_device_copy"] = round(head["roofline"]["achieved"] / out["device"]["copy_GBps"], 4)
        if not args.no_secondary:
            # the other single-GPU configurations and single ranks of both cuts of the 8-GPU tile (real local shapes, every face through the real
            # transport's self send / receive), EACH IN A FRESH PROCESS: a process that has allocated and freed lattice-sized arrays a few times draws worse
            # physical placements for the next one (the same block 1.74 ms fresh, 1.94-1.98 ms as the third lattice of a process, profiles/r03d/e), and a
            # rank's RCCL connections have to exist before its lattice is allocated (24-40 % otherwise, profiles/r01g_halo_chain.md)
            import subprocess
            sec = {}
            headline_key = next((k for k, v in SINGLE_BLOCKS.items() if v == (args.workload, args.dtype, args.coriolis, args.thermal, args.urban)), None)
            for key, flag in [(k, "--secondary-block") for k in SINGLE_BLOCKS] + [(k, "--rank-shape-block") for k in RANK_SHAPE_BLOCKS]:
                if key == headline_key and not args.size:
                    continue
                try:
                    r = subprocess.run(
                        [sys.executable, os.path.abspath(__file__), flag, key, "--steps", str(SECONDARY_STEPS), "--warmup", str(SECONDARY_WARMUP)],
                                       capture_output=True, text=True, timeout=600, env=dict(os.environ, LOCAL_RANK=str(local_rank)))
                    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    sec[key] = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": ("exit %d: " % r.returncode) + r.stderr[-300:]}
                except Exception as e:      # a secondary block never takes the headline down; its absence is visible
                    sec[key] = {"error": str(e)[:300]}
                sys.stderr.write("bench.py: secondary block %s: %s\n" % (key, sec[key].get("ms_per_step", sec[key].get("error")))); sys.stderr.flush()
                # FP16C blocks: the same block once more with the native-arithmetic kernels (LUW_OPT_NATIVE_ARITH), again in a fresh process; the block's own
                # numbers are the bit-exact kernels' ("arith": "exact"), the twin sits under "native"
                if key in NATIVE_TWINS and "error" not in sec[key]:
                    try:
                        r = subprocess.run(
                            [sys.executable, os.path.abspath(__file__), flag, key, "--arith", "native", "--steps", str(SECONDARY_STEPS), "--warmup",
                            str(SECONDARY_WARMUP)], capture_output=True, text=True, timeout=600, env=dict(os.environ, LOCAL_RANK=str(local_rank)))
                        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                        nat = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": ("exit %d: " % r.returncode) + r.stderr[-300:]}
                    except Exception as e:
                        nat = {"error": str(e)[:300]}
                    sec[key]["native"] = {k: nat[k] for k in ("value", "ms_per_step", "arith", "kernel_ms", "shell_ms", "exchange_ms", "error") if k in nat}
                    if "roofline" in nat:
                        sec[key]["native"]["roofline"] = {k: nat["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "kernel_frac")
                            if k in nat["roofline"]}
                # rank shapes: the same rank once more with its faces written in place (the one-process host's peer stores) instead of RCCL self send / receive
                if flag == "--rank-shape-block" and "error" not in sec[key]:
                    try:
                        r = subprocess.run([sys.executable, os.path.abspath(__file__), flag, key, "--rank-transport", "peer-loopback", "--steps",
                            str(SECONDARY_STEPS), "--warmup", str(SECONDARY_WARMUP)], capture_output=True, text=True, timeout=600,
                            env=dict(os.environ, LOCAL_RANK=str(local_rank)))
                        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                        tw = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": ("exit %d: " % r.returncode) + r.stderr[-300:]}
                    except Exception as e:
                        tw = {"error": str(e)[:300]}
                    sec[key]["peer_loopback"] = {k: tw[k] for k in ("value", "ms_per_step", "transport", "kernel_ms", "shell_ms", "exchange_ms", "error")
                        if k in tw}
                    if "roofline" in tw:
                        sec[key]["peer_loopback"]["roofline"] = {k: tw["roofline"][k] for k in ("achieved", "frac", "kernel_frac") if k in tw["roofline"]}
            out["secondary"] = sec
            # how this box compares: the 512^3 FP32 empty channel is the best-characterised workload of the repo (3.27-3.31 ms in ten fresh processes
            # on the boxes of profiles/r02_placement_study*.txt, 3.29-3.32 ms in every default line of round 3 but one); whole boxes run everything 10-20 %
            # slower at times (profiles/r02_skew_study.md, r03g_bench_default.json) without any clock or copy-rate reading showing it
            ref_ms = 3.30
            probe = sec.get("c2_f32", {}).get("ms_per_step") if args.workload != "c2" or args.dtype != "f32" else head["ms_per_step"]
            if probe:
                out["device"]["step_probe"] = {
                    "workload": "512x512x512 FP32 empty channel (secondary c2_f32)", "ms_per_step": probe, "typical_ms_per_step": ref_ms,
                    "box_factor": round(probe / ref_ms, 3),
                    "note": "box_factor > 1.05: this box (or this session on it) runs the step kernels that much slower than the boxes the repo's "
                            "numbers were taken on; it scales every block of this line alike"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            try:
                out["parity"] = reference_parity()
            except Exception as e:      # never let the side measurement break the benchmark line
                out["parity"] = {"error": str(e)[:200]}
        sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(out)); sys.stdout.flush()
        os.dup2(2, 1)
        import torch.distributed as dist
        if dist.is_initialized():               # the one-rank world of the rank-shape blocks
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- N > 1: one process per GPU, halos over RCCL
    run_distributed(args, torch, luw, capi, kern, rank, world, local_rank, fp16c, METRIC, saved_stdout)


# ======================================================================== N > 1
PARITY_STEPS = 8                 # both time parities, eight exchanges per split axis
# zones thinner than a rank's 64-cell block: only face-owning domains feel them (FX/kernel.cpp:1537-1541,1598)
PARITY_NUDGE_CELLS, PARITY_SPONGE_CELLS = 20, 24


def parity_tile(world, D):
    """global lattice of the small urban tile of the N > 1 self-check: 64 cells per rank in y and z, in x 384 per rank where x is split
    (two shell slabs -- 64 cells wide for FP32, 128 for FP16C -- and an interior of at least 128 cells between them: rows wide enough for the
    FP16C pair kernel everywhere) and 512 where it is whole"""
    return ((384 * D[0]) if D[0] > 1 else 512, 64 * D[1], 64 * D[2])


def parity_forcing():
    nud, spg = tile_forcing()
    return dict(nud, n_cells=PARITY_NUDGE_CELLS), dict(spg, n_cells=PARITY_SPONGE_CELLS)


def owned_digests(lay, u, rho, fi, fp16c):
    """digests of a rank's OWNED cells: rho, u (3 components) and the 19 stored DDF planes (local arrays incl. halos, reference layout).
    The two zeros are one value (+-0.0; FP16C codes 0x0000 / 0x8000) and hash alike."""
    import hashlib
    own = tuple(slice(h, n - h) for h, n in zip(lay.H, lay.lN))[::-1]                  # (z, y, x)
    cut = lambda a, c: np.asarray(a).reshape((c,) + tuple(lay.lN[::-1]))[(slice(None),) + own]
    out = {}
    for name, arr, comps in (("rho", rho, 1), ("u", u, 3), ("fi", fi, 19)):
        a = np.ascontiguousarray(cut(arr, comps))
        a = np.where(a == 0x8000, 0, a).astype(a.dtype) if a.dtype == np.uint16 else a + np.float32(0.0)
        out[name] = hashlib.blake2b(a.tobytes(), digest_size=16).hexdigest()
    out["max_abs_uy"] = float(np.abs(cut(u, 3)[1]).max())
    return out


def oracle_tile(gN, fp16c, coriolis, steps, forcing=None):
    """the CHECKER: the CPU oracle on the UNDIVIDED small tile (rank 0 only)"""
    from oracle import oracle
    nud, spg = forcing or parity_forcing()
    o = oracle.OracleLBM(*gN, NU, fp16c=fp16c)
    fill_channel(o.flags, o.u, o.rho, *gN, buildings=True)
    o.set_buffer_nudging(nud["n_cells"], nud["inv_tau"], nud["downstream_face"], nud["nudge_vertical"]); o.set_top_sponge(spg["n_cells"], spg["inv_tau"])
    if coriolis:
        o.set_coriolis(*coriolis_omega())
    o.run(steps)
    return o


def oracle_digests(o, gN, D, world, fp16c):
    from latticeurbanwind_amd.distributed import DomainLayout
    out = []
    for r in range(world):
        lay = DomainLayout(gN, D, r)
        b = tuple(g // d for g, d in zip(gN, D)); c0 = tuple(c * bb for c, bb in zip(lay.coord, b))
        sl = (slice(None), slice(c0[2], c0[2] + b[2]), slice(c0[1], c0[1] + b[1]), slice(c0[0], c0[0] + b[0]))
        blk = lambda a, c: a.reshape(c, gN[2], gN[1], gN[0])[sl]
        flat = type("L", (), {"H": (0, 0, 0), "lN": b})()                                  # the block itself, no halos
        out.append(owned_digests(flat, blk(o.u, 3), blk(o.rho, 1), blk(o.fi, 19), fp16c))
    return out


BLOCK_TIMEOUT_S = int(os.environ.get("LUW_BENCH_BLOCK_TIMEOUT", "900"))     # a secondary block of the N > 1 line that does not return


def inject_failure(where):
    """test hook (tests/test_gpu_bench_distributed.py): LUW_BENCH_INJECT=<block>:raise | <block>:hang makes that block of the N > 1 line fail the way a
    first contact with real multi-GPU hardware might"""
    spec = os.environ.get("LUW_BENCH_INJECT", "")
    if spec.startswith(where + ":"):
        if spec.endswith(":hang"):
            time.sleep(10 ** 6)
        raise RuntimeError("injected failure in block %s" % where)


class LineKeeper:
    """rank 0's guarantee of ONE parseable line: holds the most complete line so far; emit() prints it once; a block that does not return within its limit
    makes the timer print the held line -- with an error in the pending block -- and end the process with a non-zero code (the launcher takes the other
    ranks down), never a re-exec."""

    def __init__(self, stdout_fd):
        import threading
        self.fd, self.lock, self.line, self.pending, self.timer, self.done = stdout_fd, threading.Lock(), None, None, None, False

    def hold(self, line, pending=None):
        with self.lock:
            self.line, self.pending = line, pending

    def emit(self, line=None):
        with self.lock:
            if self.done:
                return
            self.done = True
            os.write(self.fd, (json.dumps(line if line is not None else self.line) + "\n").encode())

    def arm(self, seconds):
        import threading

        def fire():
            line = dict(self.line or {})
            sec = dict(line.get("secondary", {}))
            sec[self.pending or "block"] = {"error": "no result within %d s: the line is printed without this block" % seconds}
            line["secondary"] = sec
            self.emit(line)
            os._exit(4)
        self.timer = threading.Timer(seconds, fire); self.timer.daemon = True; self.timer.start()

    def disarm(self):
        if self.timer:
            self.timer.cancel(); self.timer = None


def run_distributed(args, torch, luw, capi, kern, rank, world, local_rank, fp16c, METRIC, saved_stdout):
    import torch.distributed as dist
    from latticeurbanwind_amd.distributed import DomainDecomposedLBM, choose_decomposition, tile_lattice, init_rccl_process_group
    if world == 1:      # --force-distributed without a launcher: a one-rank world over loopback
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29537"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
    shared = args.share_device is not None
    if shared: dist.init_process_group("gloo")
    else: init_rccl_process_group(local_rank)
    # a CPU-side group for waiting: while child processes of rank 0 measure the one-process host on every device, the ranks must not sit in an RCCL barrier (a
    # kernel spinning on their GPUs, where rank 0's domains are running)
    side = None if shared else dist.new_group(backend="gloo")
    os.environ.setdefault("LUW_MEASURE_WIRE", "10")                # TorchDistTransport.warm_up times the bare face exchange of every split axis
    urban = not args.no_buildings
    nud, spg = tile_forcing() if urban else (None, None)
    # this rank's GPU by itself (before the lattice exists): the slowest box sets the pace of a step
    box = device_context(torch, local_rank)
    dev_of = (lambda r: args.share_device) if shared else (lambda r: r)   # one node: rank r drives device r
    transport = "gloo + host staging (--share-device test aid: NOT a multi-GPU result)" if shared else "RCCL p2p (batch_isend_irecv)"

    def topology(lay):
        """where this rank's GPU sits and how it reaches the GPUs of its halo neighbours (HIP runtime's view of the links)"""
        info = capi.device_info(local_rank)
        links = {}
        for a in lay.split_axes():
            for sign, name in ((+1, "+"), (-1, "-")):
                nb = lay.neighbor(a, sign)
                try:
                    links["xyz"[a] + name] = dict(capi.p2p_info(local_rank, dev_of(nb)), rank=nb)
                except Exception as e:
                    links["xyz"[a] + name] = {"rank": nb, "error": str(e)[:80]}
        return {"pci_bus_id": info["pci_bus_id"], "links": links}

    # ---- self-check through the REAL transport before anything is timed
    def parity_case(D, p_fp16c, p_cor):
        gN = parity_tile(world, D)
        pn, ps = parity_forcing()
        sim = DomainDecomposedLBM(gN, D, NU, fp16c=p_fp16c, kernel=kern, device=local_rank, buffer_nudging=pn, top_sponge=ps)
        try:
            lb = sim.backend.lbm
            fill_channel(lb.flags.data, lb.u.data, lb.rho.data, sim.lNx, sim.lNy, sim.lNz, *sim.global_offset, *gN, buildings=True)
            if p_cor:
                sim.backend.set_coriolis(*coriolis_omega())
            sim.initialize(); sim.run(PARITY_STEPS)
            u, rho = sim.fields()
            mine = owned_digests(sim.layout, u, rho, lb.download_fi(), p_fp16c)
            mine["overlap"] = bool(sim.overlap)
        finally:
            sim.backend.close()
        got = [None] * world
        dist.all_gather_object(got, mine)
        case = {"dtype": "fp16c" if p_fp16c else "f32", "coriolis": bool(p_cor), "lattice": list(gN), "n_gpu": list(D), "steps": PARITY_STEPS,
                "forcing": "building array, buffer nudging %d cells, top sponge %d layers" % (PARITY_NUDGE_CELLS, PARITY_SPONGE_CELLS),
                "schedule": "shell / interior overlap, pipelined steps" if mine["overlap"] else "whole box, then exchange"}
        ora = None
        if rank == 0:
            ora = oracle_tile(gN, p_fp16c, p_cor, PARITY_STEPS)
            want = oracle_digests(ora, gN, D, world, p_fp16c)
            bad = [{"rank": r, "fields": [f for f in ("rho", "u", "fi") if got[r][f] != want[r][f]]} for r in range(world)]
            bad = [b for b in bad if b["fields"]]
            case.update(equal=not bad, mismatches=bad, cells_compared=gN[0] * gN[1] * gN[2], max_abs_uy=max(g["max_abs_uy"] for g in got),
                        compared="rho, u and the 19 stored DDF planes of every rank's owned cells (128-bit digests, all-gathered) against the CPU oracle on "
                            "the undivided lattice")
            case["equal"] = case["equal"] and case["max_abs_uy"] > 0.0           # a flow that never left the inflow profile proves nothing
        return case, ora

    D = tuple(args.n_gpu) if args.n_gpu else choose_decomposition(world, split_x=True)    # BASELINE configs[3]: the deck's literal n_gpu (8 GPUs: [4,2,1])
    if D[0] * D[1] * D[2] != world:
        raise SystemExit("bench.py: --n-gpu %s does not match %d ranks" % (D, world))
    Dalt = choose_decomposition(world)                              # x kept whole (8 GPUs: [1,4,2]): the same tile, the cut with whole rows
    cuts = [D] + ([Dalt] if (not args.n_gpu and not args.no_secondary and world > 1 and Dalt != D) else [])
    parity = {"transport": transport, "cases": []}
    if not args.no_parity:
        for Dc in cuts:
            for p_fp16c, p_cor in ((False, False), (True, True)):
                case, _ = parity_case(Dc, p_fp16c, p_cor)
                parity["cases"].append(case)
        verdict = [all(c.get("equal") for c in parity["cases"])] if rank == 0 else [None]
        dist.broadcast_object_list(verdict, src=0)
        parity["ok"] = bool(verdict[0])
        if not parity["ok"]:
            if rank == 0:
                sys.stdout.flush(); os.dup2(saved_stdout, 1)
                print(json.dumps({"metric": METRIC, "value": None, "unit": "MLUPS", "n_gpus": world,
                    "error": "decomposed run differs from the oracle: nothing was timed", "parity": parity}))
                sys.stdout.flush(); os.dup2(2, 1)
            dist.barrier(); dist.destroy_process_group()
            raise SystemExit(3)

    def run_tile(D):
        """the tile cut as n_gpu = D: returns this rank's timing block; every failure (RCCL p2p included) propagates"""
        Bx, By, Bz = args.size or (512, 512, 512)
        gN = (Bx * D[0], By * D[1], Bz * D[2]) if args.size else tile_lattice(world)
        if any(g % d for g, d in zip(gN, D)):
            raise SystemExit("bench.py: lattice %s is not divisible by n_gpu %s" % (gN, D))
        kw = dict(buffer_nudging=nud, top_sponge=spg) if urban else {}
        sim = DomainDecomposedLBM(gN, D, NU, fp16c=fp16c, kernel=kern, device=local_rank, **kw)   # RCCL connections to the neighbours first, then the lattice
        try:
            ox, oy, oz = sim.global_offset
            lb = sim.backend.lbm
            fill_channel(lb.flags.data, lb.u.data, lb.rho.data, sim.lNx, sim.lNy, sim.lNz, ox, oy, oz, *gN, buildings=urban)
            if args.coriolis:
                sim.backend.set_coriolis(*coriolis_omega())
            sim.initialize()
            sim.run(args.warmup)
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            tm = sim.run(args.steps, timed=True)
            torch.cuda.synchronize(); dist.barrier()
            dt = time.perf_counter() - t0
            tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else "cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            b = sim.layout.interior_box() if sim.overlap else sim.layout.whole_box()
            elem = 2 if fp16c else 4
            halo_out = sum(2 * 5 * lb.area(a) * elem for a in sim.layout.split_axes())      # bytes this rank sends per step (as many arrive)
            mine = {"rank": rank, "device": local_rank, "coord": list(sim.layout.coord), "local_lattice": list(sim.layout.lN),
                "wall_ms_per_step": round(dt / args.steps * 1e3, 4),
                    "kernel_ms": round(tm["kernel_ms"], 4), "kernel_cells": (b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]),
                    "shell_ms": None if tm.get("shell_ms") is None else round(tm["shell_ms"], 4), "exchange_ms": None if tm.get("exchange_ms") is None
                        else round(tm["exchange_ms"], 4),
                    "halo_bytes_out_per_step": halo_out,
                    # pack + wire + unpack + waiting for the neighbours
                    "exchange_GBps_out": round(halo_out / (tm["exchange_ms"] * 1e-3) / 1e9, 2) if tm.get("exchange_ms") else None,
                    # the bare face exchange per split axis, measured before the lattice existed
                    "wire": sim.wire,
                    "device_copy_GBps": box.get("copy_GBps"), "mclk": box.get("mclk"), "fclk": box.get("fclk")}
            mine.update(topology(sim.layout))
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            return {"D": D, "gN": gN, "dt": float(tmax.item()), "per_rank": per_rank, "overlap": sim.overlap, "one_phase": bool(sim.one_phase),
                "block": (gN[0] // D[0], gN[1] // D[1], gN[2] // D[2])}
        finally:
            sim.backend.close()

    def build_line(res, alt, alt_error, group_host):
        def block(r):
            cells = r["gN"][0] * r["gN"][1] * r["gN"][2]
            mlups = cells * args.steps / r["dt"] / 1e6
            bpl = BYTES_PER_LUP[args.dtype]
            k0 = r["per_rank"][0]
            achieved = k0["kernel_cells"] * bpl / (k0["kernel_ms"] * 1e-3) / 1e9 if k0["kernel_ms"] else None
            return mlups, bpl, achieved
        mlups, bpl, achieved = block(res)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        per_gpu = res["block"][0] * res["block"][1] * res["block"][2]
        out = {
            "metric": METRIC, "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(res["dt"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if not fp16c else "fp16c-storage/f32-arithmetic", "data": "synthetic",
            "config": {"workload": "%dx%dx%d D3Q19 %s (8 GPUs: BASELINE configs[3]) cut as n_gpu=%s, %dx%dx%d = %.0f M cells per GPU (the N = 1 line runs "
                "configs[2], 1024x1024x256 = 268 M cells on its GPU; its secondary block tile512_urban is this tile's N = 1 point), log-law profile inflow on "
                "TYPE_E faces, solid ground, SRT+Smagorinsky LES, %s DDFs%s, rho/u written by the last step only"
                       % (*res["gN"], "urban tile: building array + buffer nudging (160 m / 300 s) + top sponge (200 m / 120 s)" if urban else "channel tile",
                           list(res["D"]), *res["block"], per_gpu / 1e6,
                          "FP16C" if fp16c else "FP32", " + Coriolis force" if args.coriolis else ""),
                       "global_lattice": list(res["gN"]), "n_gpu": list(res["D"]), "cells_per_gpu": per_gpu,
                       "halo_exchange": transport + (", overlapped with the interior" if res["overlap"] else " after the whole-box kernel")
                           + (", one batch per step (faces of all axes + the 12 edge populations)" if res.get("one_phase") else ", three phases x, y, z"),
                           "kernel": args.kernel, "bytes_per_lup": bpl,
                       "rccl_version": rccl, "ranks_in_communicator": dist.get_world_size()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4) if achieved else None, "traffic": None,
                         "kernel_ms": res["per_rank"][0]["kernel_ms"],
                         "whole_job_frac": round(mlups * 1e6 * bpl / 1e9 / (HBM_PEAK_GBPS * world), 4),   # wall-clock MLUPS of all GPUs x B/LUP over N x peak
                         "note": "achieved = %g B/LUP x %d cells / mean duration of rank 0's %s kernel (HIP events on its launch stream); solid cells are "
                             "charged like fluid ones here (< 1 %% of the tile)"
                                 % (bpl, res["per_rank"][0]["kernel_cells"],
                                     "interior-box (its boundary shell and the halo exchange run concurrently on the communication stream)" if res["overlap"]
                                     else "whole-box")},
            "parity": parity if not args.no_parity else {"skipped": "--no-parity"},
            "per_rank": res["per_rank"],
        }
        sec = {}
        if alt is not None:
            m2, _, a2 = block(alt)
            sec["x_whole_n_gpu"] = {"value": round(m2, 1), "unit": "MLUPS", "ms_per_step": round(alt["dt"] / args.steps * 1e3, 4), "n_gpu": list(alt["D"]),
                "global_lattice": list(alt["gN"]),
                                    "what": "the same tile cut with x kept whole (rows stay complete memory lines; whole-row y/z shells)",
                                    "halo_exchange": transport + (", overlapped with the interior" if alt["overlap"] else " after the whole-box kernel"),
                                    "roofline_frac_rank0_kernel": round(a2 / HBM_PEAK_GBPS, 4) if a2 else None, "per_rank": alt["per_rank"]}
        elif alt_error is not None:
            sec["x_whole_n_gpu"] = {"error": alt_error}
        if group_host is not None:
            sec["group_host"] = group_host
        if sec:
            out["secondary"] = sec
        return out
    res = run_tile(D)
    # From here on rank 0 HOLDS a complete line (the headline measurement): whatever happens to a later block -- an exception on some rank, a collective that
    # never returns on hardware nobody has met -- the line is printed, once, with what did run and an `error` in the block that did not.
    keeper = LineKeeper(saved_stdout) if rank == 0 else None
    alt = alt_error = None
    if len(cuts) > 1:
        if keeper:
            keeper.hold(build_line(res, None, None, None), pending="x_whole_n_gpu")
            keeper.arm(BLOCK_TIMEOUT_S)
        try:
            inject_failure("alt")
            alt = run_tile(cuts[1])
        except Exception as e:
            alt_error = "%s: %s" % (type(e).__name__, str(e)[:300])
        finally:
            if keeper: keeper.disarm()

    # ---- the product's OTHER multi-GPU host: one process, all GPUs (luw_group_*, what luw_driver runs for decks with n_gpu > 1).  Every rank
    # has destroyed its solver; child processes of rank 0 drive all devices, one variant each, while the ranks wait in the barrier below.
    group_host = None
    if rank == 0 and not args.no_group_host and world > 1:
        keeper.hold(build_line(res, alt, alt_error, None), pending="group_host")
        try:
            inject_failure("group_host")
            group_host = run_group_host(args, D, res["gN"], [dev_of(r) for r in range(world)])
        except Exception as e:      # never takes the RCCL line down; its absence is visible
            group_host = {"error": str(e)[:300]}
    if not shared:
        torch.cuda.synchronize()
    if keeper:                                                      # a rank that never arrives (stuck in a collective of a failed block) must not cost the line
        keeper.hold(build_line(res, alt, alt_error, group_host), pending="final_barrier"); keeper.arm(BLOCK_TIMEOUT_S)
    dist.barrier(group=side)                                        # on the CPU: the ranks wait here while rank 0's child processes drive all devices
    if keeper:
        keeper.disarm()
        keeper.emit(build_line(res, alt, alt_error, group_host))
    if alt_error is not None:                 # some rank failed inside a collective sequence: the ranks are not in step any more
        os._exit(3)
    dist.barrier()
    dist.destroy_process_group()


GROUP_HOST_VARIANTS = {"peer": {"LUW_GROUP_TRANSPORT": "peer", "LUW_GROUP_THREADS": "0"},
    "peer_threads": {"LUW_GROUP_TRANSPORT": "peer", "LUW_GROUP_THREADS": "1"},
                       "rccl": {"LUW_GROUP_TRANSPORT": "rccl", "LUW_GROUP_THREADS": "0"}}
GROUP_HOST_TIMEOUT_S = int(os.environ.get("LUW_BENCH_GROUP_HOST_TIMEOUT", "420"))     # per variant


def run_group_host(args, D, gN, devices):
    """The one-process multi-domain host (csrc/luw_group.hpp behind luw_group_*: the reference's `LBM lbm(N, Dx, Dy, Dz, ...)`, what luw_driver runs
    for decks with n_gpu > 1) on the SAME tile and cut, one process over all devices, once per transport: peer stores over xGMI (one host thread, then one
    per domain) and grouped ncclSend / ncclRecv.  Each variant runs in its OWN child process under a time limit (`--group-host-child`, below): a
    host that hangs on hardware it has not met costs its own block, not the RCCL line this process still has to print."""
    import subprocess
    out = {"what": "one process drives all %d devices (luw_group_*, the deck driver's multi-GPU host); same tile and cut as the headline; each variant in a "
        "fresh child process" % len(devices),
           "devices": devices, "n_gpu": list(D), "global_lattice": list(gN)}
    env = {k: v for k, v in os.environ.items()
        if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
        and not k.startswith("TORCHELASTIC")}
    for label in GROUP_HOST_VARIANTS:
        cmd = [sys.executable, os.path.abspath(__file__), "--group-host-child", label, "--devices", ",".join(str(d) for d in devices), "--n-gpu",
            *(str(d) for d in D),
               "--global-lattice", *(str(g) for g in gN), "--dtype", args.dtype, "--kernel", args.kernel, "--steps", str(min(args.steps, 60)), "--warmup",
                   str(min(args.warmup, 5))]
        cmd += (["--coriolis"] if args.coriolis else []) + (["--no-buildings"] if args.no_buildings else []) + (["--no-parity"] if args.no_parity else [])
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=GROUP_HOST_TIMEOUT_S, env=env)     # a child past its limit is killed by its PID
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            blk = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": ("exit %d: " % r.returncode) + r.stderr[-300:]}
        except subprocess.TimeoutExpired:
            blk = {"error": "no result within %d s: child process killed" % GROUP_HOST_TIMEOUT_S}
        except Exception as e:
            blk = {"error": str(e)[:300]}
        blk["process_wall_s"] = round(time.perf_counter() - t0, 1)
        out[label] = blk
    return out


def group_host_child(args, luw, capi, kern, fp16c):
    """ONE variant of the one-process host, in this fresh process: its own self-check first (the small urban tile against the CPU oracle's undivided
    run, FP32, rho and u of every cell), then the timed tile."""
    from concurrent.futures import ThreadPoolExecutor
    label = args.group_host_child
    os.environ.update(GROUP_HOST_VARIANTS[label]); capi.reload_tuning()
    devices = [int(d) for d in args.devices.split(",")]
    D, gN, n = tuple(args.n_gpu), tuple(args.global_lattice), len(devices)
    urban = not args.no_buildings
    nud, spg = tile_forcing() if urban else (None, None)
    blk = {}
    if not args.no_parity:
        pg = parity_tile(n, D); pn, ps = parity_forcing()
        ora = oracle_tile(pg, False, False, PARITY_STEPS)
        g = luw.LBMGroup(*pg, *D, NU, devices=devices, kernel=kern, buffer_nudging=pn, top_sponge=ps)
        try:
            fill_channel(g.flags, g.u, g.rho, *pg, buildings=True)
            g.run(0); g.run(PARITY_STEPS); g.read_from_device()
            blk["parity"] = {"equal": bool(np.array_equal(g.rho, ora.rho) and np.array_equal(g.u, ora.u)), "lattice": list(pg), "steps": PARITY_STEPS,
                "dtype": "f32",
                             "compared": "rho, u of every cell against the CPU oracle on the undivided lattice"}
            blk["transport"] = capi.TRANSPORT_NAMES.get(g.transport()); blk["overlap"] = g.overlaps()
        finally:
            g.close()
        if not blk["parity"]["equal"]:
            return blk                                           # a host that computes something else is not timed
    kw = dict(buffer_nudging=nud, top_sponge=spg) if urban else {}
    g = luw.LBMGroup(*gN, *D, NU, fp16c=fp16c, devices=devices, kernel=kern, global_arrays=False, **kw)
    try:
        def fill(d):
            lN, off, _ = g.domain_info(d)
            fl, u, rho = g.domain_host(d)
            fill_channel(fl, u, rho, *lN, *off, *gN, buildings=urban)
        with ThreadPoolExecutor(max_workers=min(n, 8)) as ex:
            list(ex.map(fill, range(n)))
        if args.coriolis:
            g.set_coriolis(*coriolis_omega())
        g.initialize_from_domains()
        g.run(args.warmup)
        t0 = time.perf_counter()
        kms = g.run_timed(args.steps)
        dt = time.perf_counter() - t0
        cells = gN[0] * gN[1] * gN[2]
        blk.update(value=round(cells * args.steps / dt / 1e6, 1), unit="MLUPS", ms_per_step=round(dt / args.steps * 1e3, 4), steps=args.steps,
            warmup=args.warmup, domain0_kernel_ms=round(kms, 4),
                   transport=capi.TRANSPORT_NAMES.get(g.transport()), direct_peer_stores=g.direct_peer_stores(), overlap=g.overlaps(),
                   host_threads="one per domain" if GROUP_HOST_VARIANTS[label]["LUW_GROUP_THREADS"] == "1" else "one")
    finally:
        g.close()
    return blk


if __name__ == "__main__":
    main()
