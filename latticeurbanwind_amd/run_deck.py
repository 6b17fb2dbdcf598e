"""Decomposed deck run: `n_gpu = [Dx, Dy, Dz]` decks on Dx*Dy*Dz GPUs of one node, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m latticeurbanwind_amd.run_deck case/conf.luw

The host stage is the C++ driver's (latticeurbanwind_amd/host/luw_driver --export-setup: deck, sizing, units, STL + device
voxeliser, boundary builders, VK tables), run once by rank 0; every rank then cuts its block out of the exported state, and
the run loop of the reference (FX/setup.cpp:4117-4911: unsteady u outputs, purge_avg window on the device, final u / rho,
_avg VTK with tke / TI / TLS) is driven here over DomainDecomposedLBM (RCCL halos).  Blocks are assembled through files in
the project's proj_temp (ranks share the node's file system), rank 0 writes the same VTK files as the single-GPU driver.
With one rank it reproduces the C++ driver's files (tests/test_gpu_run_deck.py).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL between the ranks' processes (normally already exported)

from . import capi
from .distributed import DomainDecomposedLBM, DomainLayout, init_rccl_process_group

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "host", "luw_driver")
f32 = np.float32


def _f(bits):
    return np.array([bits], np.uint32).view(np.float32)[0]


class _Single:
    """stand-in for torch.distributed when the module is started without a launcher (one GPU)"""
    rank, world = 0, 1
    def barrier(self): pass


class _Dist:
    def __init__(self, dist):
        self.d = dist; self.rank = dist.get_rank(); self.world = dist.get_world_size()
    def barrier(self): self.d.barrier()


# ----------------------------------------------------------------------------- VTK (FX/lbm.hpp:307-356, FX/setup.cpp:2513-2683)
def vtk_header(filename, m, Nz_out):
    Nx, Ny, _ = m["N"]
    return ("# vtk DataFile Version 3.0\nFluidX3D %s\nBINARY\nDATASET STRUCTURED_POINTS\nDIMENSIONS %d %d %d\nORIGIN %s\nSPACING %s\nPOINT_DATA %d\n"
            % (os.path.basename(filename), Nx, Ny, Nz_out, m["output"]["vtk_origin"], m["output"]["vtk_spacing"], Nx * Ny * Nz_out)).encode()


def write_field_vtk(filename, m, soa, comps, factor, offset=None):
    """soa: (comps, Nz, Ny, Nx) float32 in lattice units -> AoS big-endian floats in SI units, first Nz_out layers; with an
    offset the value is data*factor+offset (the T field in Kelvin, FX/lbm.hpp:343), else factor*data"""
    os.makedirs(os.path.dirname(filename), exist_ok=True)
    Nz_out = m["Nz_out"]
    with open(filename, "wb") as f:
        f.write(vtk_header(filename, m, Nz_out))
        f.write(("SCALARS data float %d\nLOOKUP_TABLE default\n" % comps).encode())
        for z in range(Nz_out):      # layer by layer: bounded memory on billion-cell lattices
            lay = (np.asarray(soa[:, z]) * f32(factor) + f32(offset)).astype(np.float32) if offset is not None else (f32(factor) * np.asarray(soa[:, z])).astype(np.float32)
            f.write(np.ascontiguousarray(np.moveaxis(lay, 0, -1)).astype(">f4").tobytes())


def write_avg_vtk(filename, m, avg_u, avg_rho, m2, count, solid, si_u, si_rho, spacing, avg_T=None, T_factor=1.0, T_offset=0.0):
    """avg_u (Nz,Ny,Nx,3), avg_rho / m2[k] (Nz,Ny,Nx), all lattice units, already cut to Nz_out layers"""
    os.makedirs(os.path.dirname(filename), exist_ok=True)
    Nzo, Ny, Nx = avg_rho.shape
    uf, rf = f32(si_u), f32(si_rho)
    out = m["output"]
    with open(filename, "wb") as f:
        f.write(vtk_header(filename, m, Nzo))
        def field(name, a, comps, factor):
            f.write(("SCALARS %s float %d\nLOOKUP_TABLE default\n" % (name, comps)).encode())
            f.write(((a * f32(factor)) + f32(0.0)).astype(">f4").tobytes())
        field("u_avg", avg_u, 3, uf)
        field("rho_avg", avg_rho, 1, rf)
        if avg_T is not None:
            f.write(b"SCALARS T_avg float 1\nLOOKUP_TABLE default\n")
            f.write((avg_T * f32(T_factor) + f32(T_offset)).astype(">f4").tobytes())
        fluid = np.where(solid, f32(0), f32(1)).astype(np.float32)
        zero = np.zeros(avg_rho.shape, np.float32)
        tke, ti, tls = zero.copy(), zero.copy(), zero.copy()
        if count > 1 and (out["tke"] or out["ti"] or out["tls"]):
            inv_n = f32(1.0) / f32(count)
            var = [np.maximum(m2[k] * inv_n, f32(0)) for k in range(3)]
            var_sum = (var[0] + var[1]) + var[2]
            live = ~solid
            if out["tke"]:
                tke = np.where(live, f32(0.5) * var_sum, f32(0)).astype(np.float32)
            if out["ti"]:
                ax, ay, az = avg_u[..., 0], avg_u[..., 1], avg_u[..., 2]
                umag = np.sqrt((ax * ax + ay * ay) + az * az)
                with np.errstate(divide="ignore", invalid="ignore"):
                    ti = np.where(live & (umag > f32(1.0e-9)) & (var_sum > 0), np.sqrt(var_sum * (f32(1.0) / f32(3.0))) / umag, f32(0)).astype(np.float32)
            if out["tls"]:
                dx = np.maximum(f32(spacing), f32(1.0e-12))
                su = (avg_u * uf).astype(np.float32)
                def grad(axis, n):        # one-sided at the ends, central inside; axis: 0 z, 1 y, 2 x of the (z,y,x,3) array
                    ip = np.minimum(np.arange(n) + 1, n - 1); im = np.maximum(np.arange(n) - 1, 0)
                    inv = np.where(ip > im, f32(1.0) / ((ip - im).astype(np.float32) * dx), f32(0)).astype(np.float32)
                    shape = [1, 1, 1, 1]; shape[axis] = n
                    return ((np.take(su, ip, axis) - np.take(su, im, axis)) * inv.reshape(shape)).astype(np.float32)
                gx, gy, gz = grad(2, Nx), grad(1, Ny), grad(0, Nzo)
                duxdx, duydx, duzdx = gx[..., 0], gx[..., 1], gx[..., 2]
                duxdy, duydy, duzdy = gy[..., 0], gy[..., 1], gy[..., 2]
                duxdz, duydz, duzdz = gz[..., 0], gz[..., 1], gz[..., 2]
                Sxy = f32(0.5) * (duxdy + duydx); Sxz = f32(0.5) * (duxdz + duzdx); Syz = f32(0.5) * (duydz + duzdy)
                S_mag = np.sqrt(np.maximum(f32(0), f32(2.0) * (((duxdx * duxdx + duydy * duydy) + duzdz * duzdz) + f32(2.0) * ((Sxy * Sxy + Sxz * Sxz) + Syz * Syz))))
                k_loc = (f32(0.5) * var_sum) * (uf * uf)
                cap = f32(max(Nx, Ny, Nzo)) * dx
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = np.where((S_mag > f32(1.0e-10)) & (k_loc > 0), np.sqrt(k_loc) / S_mag, f32(0))
                tls = np.where(live, np.minimum(np.maximum(t, f32(0)), cap), f32(0)).astype(np.float32)
        field("fluid", fluid, 1, 1.0)
        if out["tke"]: field("tke", tke, 1, uf * uf)
        if out["ti"]: field("TI", ti, 1, 1.0)
        if out["tls"]: field("TLS", tls, 1, 1.0)


# ----------------------------------------------------------------------------- one case
def _default_name(prefix, name, t):
    return "%s%s-%09d.vtk" % (prefix, name, t)


def runtime_decomposition(N, D_deck):
    """The domain grid the run uses for a deck's n_gpu: the same number of domains, but with the memory-fastest axis (x) kept
    whole whenever the lattice divides that way -- whole rows cost a rank 5-8 % over an undivided lattice, an x split 12 %+
    (DESIGN.md section 6), and a decomposed run equals the undivided one bit for bit whatever the split.  Among the x-whole
    candidates the one with the least halo area wins; the deck's own grid is the fallback."""
    world = D_deck[0] * D_deck[1] * D_deck[2]
    best, best_area = None, None
    for a in range(1, world + 1):
        if world % a: continue
        b = world // a
        if N[1] % a or N[2] % b or N[1] // a < 4 or N[2] // b < 4: continue
        area = (N[0] * (N[2] // b) if a > 1 else 0) + (N[0] * (N[1] // a) if b > 1 else 0)      # y faces + z faces per rank
        if best is None or area < best_area:
            best, best_area = (1, a, b), area
    return best if best is not None else tuple(D_deck)


def run_case(m, G, device, log, make_sim=None, literal_n_gpu=False):
    Nx, Ny, Nz = m["N"]; D_deck = tuple(m["n_gpu"]); Ncells = Nx * Ny * Nz
    if D_deck[0] * D_deck[1] * D_deck[2] != G.world:
        raise SystemExit("run_deck: the deck asks for n_gpu=%s = %d domains but %d ranks were launched" % (list(D_deck), D_deck[0] * D_deck[1] * D_deck[2], G.world))
    D = D_deck if literal_n_gpu else runtime_decomposition((Nx, Ny, Nz), D_deck)
    if D != D_deck:
        log("| Decomposition   | deck n_gpu=%s runs as %s: x rows kept whole, same results" % (list(D_deck), list(D)))
    nu = float(_f(m["nu_bits"])); si_u = _f(m["si_u_bits"]); si_rho = _f(m["si_rho_bits"]); spacing = _f(m["spacing_bits"])
    nud = m["buffer"]; spg = m["sponge"]
    kw = dict(fp16c=bool(m["fp16c"]),
              buffer_nudging=dict(n_cells=nud["n_cells"], inv_tau=float(_f(nud["inv_tau_bits"])), downstream_face=nud["downstream_face"], nudge_vertical=nud["nudge_vertical"]) if nud["active"] else None,
              top_sponge=dict(n_cells=spg["n_cells"], inv_tau=float(_f(spg["inv_tau_bits"]))) if spg["active"] else None)
    th = m.get("thermal", {"on": 0}); thermal = bool(th["on"])
    if thermal:
        kw["alpha"] = float(_f(th["alpha_bits"]))
    sim = make_sim((Nx, Ny, Nz), D, nu, G.rank, kw) if make_sim else DomainDecomposedLBM((Nx, Ny, Nz), D, nu, rank=G.rank, device=device, **kw)
    lay = sim.layout
    state = np.memmap(m["state"], np.uint8, "r")
    gflags = state[:Ncells]; gu = state[Ncells:Ncells + 12 * Ncells].view(np.float32)
    gT = state[13 * Ncells:17 * Ncells].view(np.float32) if thermal else None
    sim.set_fields_from_global(gflags, gu, None, gT)
    om = [float(_f(b)) for b in m["omega_bits"]]
    if any(om):
        sim.backend.set_coriolis(*om)
    if m["vk"]["on"]:            # von-Karman tables are global; keep the points that sit on cells this rank owns
        raw = np.fromfile(m["vk_tables"], np.uint8)
        P, M = (int(v) for v in raw[:16].view(np.uint64))
        o = 16; cell = raw[o:o + 8 * P].view(np.uint64); o += 8 * P
        face = raw[o:o + P]; o += P
        pdat = raw[o:o + 28 * P].view(np.float32).reshape(7, P); o += 28 * P       # SoA: x, y, z, ubx, uby, ubz, sigma
        mdat = raw[o:o + 200 * M].view(np.float32)
        x = (cell % Nx).astype(np.int64); y = ((cell // Nx) % Ny).astype(np.int64); z = (cell // (Nx * Ny)).astype(np.int64)
        lx, ly, lz = x - lay.O[0], y - lay.O[1], z - lay.O[2]
        own = np.ones(P, bool)
        for l, a in ((lx, 0), (ly, 1), (lz, 2)):
            lo, hi = lay.nonhalo_range(a); own &= (l >= lo) & (l < hi)
        lcell = (lx + (ly + lz * lay.lN[1]) * lay.lN[0])[own].astype(np.uint64)
        if own.any():
            sim.backend.vk_attach(lcell, face[own], np.ascontiguousarray(pdat[:, own]).ravel(), mdat, M, m["vk"]["stride"], bool(m["vk"]["interp"]))
            sim.pre_step = sim.backend.vk_apply
    st = m["steps"]; total, unsteady, avg_window, avg_stride = st["total"], st["unsteady"], st["avg_window"], max(1, st["avg_stride"])
    avg_start = total - avg_window + 1 if avg_window > 0 else None
    if avg_window > 0:
        sim.backend.stats_reset()
    # probe columns (FX/setup.cpp:4269-4395,4495-4506): every rank samples the probe cells it owns, rank 0 writes the CSVs
    pr = m.get("probes", {"columns": []}); cols = pr["columns"]; probe_start = pr.get("start_t", 0) if cols else None
    my_cells, my_slots = [], []            # slot = (column index, level index)
    for ci, col in enumerate(cols):
        lx, ly = col["x"] - lay.O[0], col["y"] - lay.O[1]
        for li, z in enumerate(col["z"]):
            l = (lx, ly, z - lay.O[2])
            if all(lay.nonhalo_range(a)[0] <= l[a] < lay.nonhalo_range(a)[1] for a in range(3)):
                my_cells.append(l[0] + (l[1] + l[2] * lay.lN[1]) * lay.lN[0]); my_slots.append((ci, li))
    if my_cells:
        sim.backend.gather_attach(np.array(my_cells, np.uint64))
    probe_times, probe_vals = [], []      # per sampled step: float32 (n_my_cells, 3) in lattice units
    out = m["output"]; raw_prefix = os.path.join(out["results_vtk_dir"], out["raw_prefix"])
    scratch = os.path.dirname(m["state"])
    sim.initialize()
    log("| Decomposed run  | n_gpu=%s, local %s, %s, %d steps" % (list(D), list(lay.lN), "shell/interior overlap" if sim.overlap else "whole box + exchange", total))
    t0 = time.perf_counter()

    def gather(name, comps, local, t):
        """every rank drops its owned block into one file; returns the (comps,Nz,Ny,Nx) memmap on rank 0"""
        path = os.path.join(scratch, "%s_%d.gather" % (name, t))
        if G.rank == 0:
            np.memmap(path, np.float32, "w+", shape=(comps, Nz, Ny, Nx)).flush()
        G.barrier()
        mm = np.memmap(path, np.float32, "r+", shape=(comps, Nz, Ny, Nx))
        blk, (x0, y0, z0) = sim.interior_to_global(local, comps)
        mm[:, z0:z0 + blk.shape[1], y0:y0 + blk.shape[2], x0:x0 + blk.shape[3]] = blk
        mm.flush(); del mm
        G.barrier()
        return np.memmap(path, np.float32, "r", shape=(comps, Nz, Ny, Nx)) if G.rank == 0 else None, path

    def write_u(t):
        u, _ = sim.fields()
        g, path = gather("u", 3, u, t)
        if G.rank == 0:
            fn = _default_name(raw_prefix, "u", t); write_field_vtk(fn, m, g, 3, si_u); log("| VTK file        | %s saved" % fn); del g; os.remove(path)
        G.barrier()

    t = 0; last_u = None
    while t < total:
        nxt = total
        if unsteady > 0:
            nxt = min(nxt, (t // unsteady + 1) * unsteady)
        if cols:
            nxt = min(nxt, max(t + 1, probe_start))       # every step of the probe window is observed
        sample = None                                     # statistics samples inside (t, nxt] ride along, no host round trip
        if avg_window > 0:
            s = max(t + 1, avg_start); off = (s - avg_start) % avg_stride
            if off: s += avg_stride - off
            if s <= nxt: sample = (s - t, avg_stride)
        sim.run(nxt - t, sample=sample); t = nxt
        if unsteady > 0 and t % unsteady == 0:
            write_u(t); last_u = t
        if cols and t >= probe_start:
            probe_times.append(t)
            if my_cells:
                probe_vals.append(sim.backend.gather_u())
    secs = time.perf_counter() - t0
    log("| Solver          | %d steps in %.3f s = %.1f MLUPs" % (total, secs, Ncells * total / secs * 1e-6))
    if last_u != t:
        write_u(t)
    u, rho = sim.fields()
    g, path = gather("rho", 1, rho, t)
    if G.rank == 0:
        fn = _default_name(raw_prefix, "rho", t); write_field_vtk(fn, m, g, 1, si_rho); log("|                 | %s saved" % fn); del g; os.remove(path)
    G.barrier()
    if thermal:      # raw_T in Kelvin, FX/setup.cpp:4771-4774
        g, path = gather("T", 1, sim.backend.download_T(), t)
        if G.rank == 0:
            fn = _default_name(raw_prefix, "T", t); write_field_vtk(fn, m, g, 1, _f(th["unit_K_bits"]), _f(th["unit_K_offset_bits"])); log("|                 | %s saved" % fn); del g; os.remove(path)
        G.barrier()
    if avg_window > 0:
        sd = sim.backend.stats_download()
        lN = lay.lN
        au = sd["avg_u"].reshape(lN[2], lN[1], lN[0], 3).transpose(3, 0, 1, 2)          # AoS -> (3,z,y,x) like a field
        parts = {}
        fields = [("avg_u", 3, au), ("avg_rho", 1, sd["avg_rho"]), ("m2u", 1, sd["m2_u"]), ("m2v", 1, sd["m2_v"]), ("m2w", 1, sd["m2_w"])]
        if thermal:
            fields.append(("avg_T", 1, sim.backend.stats_download_T()))
        for name, comps, arr in fields:
            parts[name] = gather(name, comps, np.ascontiguousarray(arr), t)
        if G.rank == 0 and sd["count"] > 0:
            Nzo = m["Nz_out"]
            A = lambda k: np.asarray(parts[k][0][:, :Nzo])
            solid = (np.asarray(gflags).reshape(Nz, Ny, Nx)[:Nzo] & 1) != 0
            fn = _default_name(os.path.join(out["results_vtk_dir"], out["avg_name"]), "", t)
            write_avg_vtk(fn, m, np.ascontiguousarray(np.moveaxis(A("avg_u"), 0, -1)), A("avg_rho")[0], [A("m2u")[0], A("m2v")[0], A("m2w")[0]], sd["count"], solid, si_u, si_rho, spacing,
                          A("avg_T")[0] if thermal else None, _f(th["si_dT_bits"]) if thermal else 1.0, _f(th["si_T0_bits"]) if thermal else 0.0)
            log("| VTK file        | %s saved" % fn); log("| Avg samples     | %d" % sd["count"])
        G.barrier()
        if G.rank == 0:
            for g, path in parts.values():
                del g
                os.remove(path)
    if cols:
        part = os.path.join(scratch, "probes_rank%d.npz" % G.rank)
        np.savez(part, slots=np.array(my_slots, np.int64).reshape(-1, 2), vals=np.array(probe_vals, np.float32).reshape(len(probe_times), len(my_cells), 3) if my_cells else np.zeros((len(probe_times), 0, 3), np.float32))
        G.barrier()
        if G.rank == 0:
            unit_m, unit_s = _f(m["unit_m_bits"]), _f(m["unit_s_bits"]); dt = float(pr["dt_si"])
            series = [np.zeros((len(probe_times), len(c["z"]), 3), np.float32) for c in cols]
            for r in range(G.world):
                d = np.load(os.path.join(scratch, "probes_rank%d.npz" % r))
                for k, (ci, li) in enumerate(d["slots"]):
                    series[ci][:, li, :] = (d["vals"][:, k, :] * unit_m) / unit_s          # Units::si_u, FP32
            fmt = lambda v: ("%.6f" % v).rstrip("0").rstrip(".") if "." in ("%.6f" % v) else ("%.6f" % v)
            fix = lambda v: "0" if fmt(v) in ("", "-0") and v == 0 else fmt(v)
            os.makedirs(pr["results_dir"], exist_ok=True)
            for c, sr in zip(cols, series):
                with open(os.path.join(pr["results_dir"], c["stem"] + ".csv"), "w") as f:
                    f.write("height (m)" + "".join("," + fix(tt * dt) for tt in probe_times) + "\n")
                    for li, hb in enumerate(c["height_bits"]):
                        f.write(fix(float(_f(hb))) + "".join(",%s:%s:%s" % tuple(fix(float(x)) for x in sr[ti, li]) for ti in range(len(probe_times))) + "\n")
            log("| Probe files     | %d CSV saved to RESULTS" % len(cols))
        G.barrier()
    sim.backend.close()
    G.barrier()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m latticeurbanwind_amd.run_deck")
    ap.add_argument("deck"); ap.add_argument("--ddf", choices=["fp32", "fp16c"], default="fp16c")
    ap.add_argument("--host-voxeliser", action="store_true", help="export the set-up with --dry-run (no GPU in the host stage)")
    ap.add_argument("--literal-n-gpu", action="store_true", help="split exactly as the deck's n_gpu says (default: same domain count, x kept whole when the lattice allows)")
    ap.add_argument("--share-device", type=int, default=None, help="test aid: all ranks use this one GPU and exchange halos through gloo + host staging")
    a = ap.parse_args(argv)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and a.share_device is not None:
        local_rank = a.share_device
    if world > 1 and a.share_device is None and torch.cuda.device_count() == 0:
        raise SystemExit("run_deck: no GPU visible; the solver has no CPU fallback")
    parent = os.path.dirname(os.path.abspath(a.deck)) or "."
    scratch = os.path.join(parent, "proj_temp", "run_deck_setup")
    log = (lambda s: print(s, flush=True)) if rank == 0 else (lambda s: None)
    # The C++ host stage runs first, on rank 0, BEFORE any rank opens its GPU or joins the process group: the other ranks wait in
    # the rendezvous of init_process_group (no GPU involved), so the host stage's own device work (voxeliser) never shares the
    # card with more processes than the run itself (several ranks on one GPU in the tests: the box allows six).
    if rank == 0:
        shutil.rmtree(scratch, ignore_errors=True)
        cmd = [DRIVER, a.deck, "--ddf", a.ddf, "--device", str(local_rank), "--export-setup", scratch] + (["--dry-run"] if a.host_voxeliser else [])
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        sys.stdout.write(r.stdout); sys.stdout.flush()
        if r.returncode != 0:
            raise SystemExit("run_deck: host stage failed (exit %d)" % r.returncode)
    if world > 1:
        import datetime
        import torch.distributed as dist
        if a.share_device is not None:
            dist.init_process_group("gloo", timeout=datetime.timedelta(hours=4))
            torch.cuda.set_device(local_rank)
        else:
            torch.cuda.set_device(local_rank)
            init_rccl_process_group(local_rank, timeout=datetime.timedelta(hours=4))
        G = _Dist(dist)
    else:
        G = _Single()
    capi.load()
    G.barrier()
    k = 1
    while os.path.exists(os.path.join(scratch, "case%d.json" % k)):
        run_case(json.load(open(os.path.join(scratch, "case%d.json" % k))), G, local_rank, log, literal_n_gpu=a.literal_n_gpu)
        k += 1
    G.barrier()
    if G.rank == 0:
        shutil.rmtree(scratch, ignore_errors=True)
    if world > 1:
        G.d.destroy_process_group()


if __name__ == "__main__":
    main()
