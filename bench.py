#!/usr/bin/env python3
"""bench.py -- MLUPS of the D3Q19 collide-stream hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): 512^3 D3Q19 channel per GPU, FP32 DDFs, Smagorinsky LES on,
z=0 plane solid, the other five outer faces TYPE_E with a log-law inflow profile, interior initialised with the
same profile, rho=1.  N>1 runs the weak-scaled tile (512^3 cells per GPU; 8 GPUs = BASELINE configs[3] 2048x1024x512, cut as
n_gpu=[1,4,2] by default, --n-gpu 4 2 1 for the deck's literal grid) with one-cell halos exchanged over RCCL.  A "step" is one stream_collide pass over the whole lattice;
rho/u are written by the last step only (153 B/LUP mode); data is synthetic and resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.
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


def loglaw_profile(nz, u_max=0.1):
    """lattice-unit inflow speed per z level: log law over the cell centres above the solid z=0 plane"""
    z = (np.arange(nz, dtype=np.float64) - 0.5) * 2.0          # metres above ground, cell = 2 m
    z0 = 0.3
    u = np.log(np.maximum(z, 0.0) / z0 + 1.0)
    u[0] = 0.0
    return (u_max * u / u.max()).astype(np.float32)


def channel_state(Nx, Ny, Nz, gx0=0, gy0=0, gz0=0, GNx=None, GNy=None, GNz=None, buildings=False):
    """flags/u/rho of the (sub)box [gx0,gx0+Nx) x ... of the global channel GNx x GNy x GNz (wind along +x).
    buildings=True adds BASELINE configs[2]'s solid mask: an array of axis-aligned boxes, footprint 24x24 cells on a 64-cell
    pitch, heights 8+((7i+13j) mod 48) cells (closed form)"""
    GNx, GNy, GNz = GNx or Nx, GNy or Ny, GNz or Nz
    prof = loglaw_profile(GNz)
    zs = (np.arange(Nz) + gz0) % GNz; ys = (np.arange(Ny) + gy0) % GNy; xs = (np.arange(Nx) + gx0) % GNx
    flags = np.zeros((Nz, Ny, Nx), np.uint8)
    bz = (zs == GNz - 1); by = (ys == 0) | (ys == GNy - 1); bx = (xs == 0) | (xs == GNx - 1)
    flags[bz, :, :] = 2; flags[:, by, :] = 2; flags[:, :, bx] = 2
    flags[zs == 0, :, :] = 1
    if buildings:
        i, j = xs // 64, ys // 64
        inx = (xs % 64 >= 20) & (xs % 64 < 44) & (xs > 0) & (xs < GNx - 1); iny = (ys % 64 >= 20) & (ys % 64 < 44) & (ys > 0) & (ys < GNy - 1)
        h = 8 + ((7 * i[None, :] + 13 * j[:, None]) % 48)                       # (Ny, Nx) building height in cells
        solid = (inx[None, :] & iny[:, None])[None, :, :] & (zs[:, None, None] >= 1) & (zs[:, None, None] < 1 + h[None, :, :])
        flags[solid] = 1
    u = np.zeros((3, Nz, Ny, Nx), np.float32)
    u[0] = prof[zs][:, None, None]
    u[0][flags == 1] = 0.0
    rho = np.ones((Nz, Ny, Nx), np.float32)
    return flags.ravel(), u.ravel(), rho.ravel()


def reference_parity():
    """u-field RMSE against the REAL reference, measured now: the deck driver runs the committed synthetic case B (48x40x24, one
    building, LES, 64 steps, FP32 DDFs) on this GPU and its final velocity file is compared with the file the reference solver
    (FluidX3D, FP32 build, run on an MI355X through OpenCL) wrote for the same deck (tests/golden/ref_fp32_CaseB.npz)."""
    import glob, shutil, subprocess, tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from vtkio import read_vtk
    drv = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_fp32_CaseB.npz"))
    tmp = tempfile.mkdtemp()
    try:
        shutil.copytree(os.path.join(ROOT, "tests", "golden", "refcases", "CaseB"), os.path.join(tmp, "CaseB"))
        r = subprocess.run([drv, os.path.join(tmp, "CaseB", "conf.luwpf"), "--ddf", "fp32"], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": "driver exit %d" % r.returncode}
        h, f = read_vtk(glob.glob(os.path.join(tmp, "CaseB", "RESULTS", "vtk", "*_raw_u-000000064.vtk"))[0])
        fac = np.float32(5.0) / np.float32(0.1)                      # si_ref_u = max profile U = 5 m/s, u_lbm = 0.1
        d = ((f["data"] - gold["u64"]) / fac)[~gold["solid"]].astype(np.float64)
        return {"u_rmse_vs_reference": float(np.sqrt((d ** 2).sum(-1).mean())), "unit": "lattice units", "steps": 64, "case": "tests/golden/refcases/CaseB (48x40x24, FP32 DDFs)",
                "tolerance": 1e-5}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def coriolis_omega():
    """Omega_earth (0, cos phi, sin phi) dt in lattice units at 31.25 deg N for cell = 2 m, U_ref = 10 m/s, u_lbm = 0.1 (FX/setup.cpp:3800-3823)"""
    import math
    dt = 2.0 * 0.1 / 10.0
    return 0.0, 7.292115e-5 * math.cos(math.radians(31.25)) * dt, 7.292115e-5 * math.sin(math.radians(31.25)) * dt


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


def cpu_baseline(max_seconds=20.0):
    """The CPU oracle (our C/OpenMP restatement of the reference kernel, kind "port") timed on this box's host
    cores on a bounded sample of the same workload recipe: 128^3 channel, FP32 DDFs, as many steps as fit.  The
    thread count is the fastest of a short sweep (over-subscription inside a CPU-limited container is disastrous)."""
    from oracle import oracle
    N = 128
    o = oracle.OracleLBM(N, N, N, 1.48e-7)
    fl, u, rho = channel_state(N, N, N)
    o.flags[:] = fl; o.u[:] = u; o.rho[:] = rho
    o.run(2)
    cores = usable_cores()
    cands = sorted({max(1, c) for c in (cores, cores // 2, cores // 4, 64, 32, 16, 8) if c <= cores}, reverse=True)
    best_t, best_rate = cands[-1], 0.0
    for t in cands:
        oracle.set_threads(t)
        o.run(1)
        t0 = time.perf_counter(); o.run(2); dt = time.perf_counter() - t0
        if 2 * N ** 3 / dt > best_rate:
            best_rate, best_t = 2 * N ** 3 / dt, t
    oracle.set_threads(best_t)
    steps, t0 = 0, time.perf_counter()
    while True:
        o.run(4); steps += 4
        dt = time.perf_counter() - t0
        if dt > max_seconds or steps >= 2000:
            break
    return {"value": round(N ** 3 * steps / dt / 1e6, 1), "unit": "MLUPS", "cores": best_t, "kind": "port",
            "sample": "%d steps of a 128^3 FP32 channel (same recipe as the GPU workload) in %.1f s, OpenMP threads swept over %s of %d usable cores" % (steps, dt, cands, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, nargs=3, default=None, help="per-GPU lattice (default 512 512 512; with N > 1 the default global lattice is the BASELINE tile, 2048x1024x512 on 8 GPUs)")
    ap.add_argument("--n-gpu", type=int, nargs=3, default=None, help="domain grid Dx Dy Dz (default: x kept whole, e.g. 1 4 2 on 8 GPUs; the deck's literal 4 2 1 is accepted)")
    ap.add_argument("--dtype", choices=["f32", "fp16c"], default="f32")
    ap.add_argument("--kernel", choices=["auto", "scalar", "scalar_cached", "scalar_nt_all", "vec4", "vec2", "vec1", "pair", "exp_copy", "exp_noshift"], default="auto", help="exp_* are measurement-only variants (no physics)")
    ap.add_argument("--buildings", action="store_true", help="BASELINE configs[2] solid mask (box array); use with --size 1024 1024 256")
    ap.add_argument("--coriolis", action="store_true", help="Coriolis body force at 31.25 deg N (BASELINE configs[4]): every cell takes the forced path")
    ap.add_argument("--thermal", action="store_true", help="also run the thermal D3Q7 lattice (the shipped reference build always does): +7 DDF planes and T")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--share-device", type=int, default=None, help="test aid: all ranks use this one GPU, halos through gloo + host staging (plumbing check of the N > 1 path on a 1-GPU box)")
    ap.add_argument("--force-distributed", action="store_true", help="take the N > 1 code path (process group, DomainDecomposedLBM) even with one rank: plumbing check")
    ap.add_argument("--every-step-fields", action="store_true", help="write rho,u every step like the reference's UPDATE_FIELDS (169 B/LUP)")
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
    torch.cuda.set_device(local_rank)
    luw.load()
    Nx, Ny, Nz = args.size or (512, 512, 512)
    kern = {"auto": capi.KERNEL_AUTO, "scalar": capi.KERNEL_SCALAR, "scalar_cached": capi.KERNEL_SCALAR_CACHED, "vec4": capi.KERNEL_VEC4, "vec2": capi.KERNEL_VEC2, "vec1": capi.KERNEL_VEC1, "pair": capi.KERNEL_PAIR, "scalar_nt_all": capi.KERNEL_SCALAR_NT_ALL, "exp_copy": capi.KERNEL_EXP_COPY, "exp_noshift": capi.KERNEL_EXP_NOSHIFT}[args.kernel]
    fp16c = args.dtype == "fp16c"
    nu = 1.48e-7                                     # units.nu(1.48e-5) for cell = 2 m, U_ref = 10 m/s at u_lbm = 0.1

    if world == 1 and not args.force_distributed:
        lbm = luw.LBM(Nx, Ny, Nz, nu, fp16c=fp16c, kernel=kern, device=local_rank, update_fields_every_step=args.every_step_fields, alpha=(2.1e-7 if args.thermal else None))
        fl, u, rho = channel_state(Nx, Ny, Nz, buildings=args.buildings)
        lbm.flags.data[:] = fl; lbm.u.data[:] = u; lbm.rho.data[:] = rho
        if args.coriolis:
            lbm.set_coriolis(*coriolis_omega())
        lbm.run(0)
        lbm.run(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms = lbm.run_timed(args.steps)        # K steps, HIP events around each launch on the launch stream
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cells = Nx * Ny * Nz
        D = (1, 1, 1)
    else:
        import torch.distributed as dist
        from latticeurbanwind_amd.distributed import DomainDecomposedLBM, choose_decomposition, tile_lattice, init_rccl_process_group
        if world == 1:      # --force-distributed without a launcher: a one-rank world over loopback
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29537"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                os.environ.setdefault(k, v)
        if args.share_device is not None: dist.init_process_group("gloo")
        else: init_rccl_process_group(local_rank)
        D = tuple(args.n_gpu) if args.n_gpu else choose_decomposition(world)
        if D[0] * D[1] * D[2] != world:
            raise SystemExit("bench.py: --n-gpu %s does not match %d ranks" % (D, world))
        # weak scaling: 512^3 cells per GPU.  Default global lattice = the BASELINE tile for this GPU count (8: 2048x1024x512)
        gN = (Nx * D[0], Ny * D[1], Nz * D[2]) if args.size else tile_lattice(world)
        if any(g % d for g, d in zip(gN, D)):
            raise SystemExit("bench.py: lattice %s is not divisible by n_gpu %s" % (gN, D))
        Nx, Ny, Nz = (g // d for g, d in zip(gN, D))          # per-GPU block (without halos)
        exchange_note = None
        gloo_group = None                                   # only used if RCCL point-to-point fails below
        if args.share_device is None:
            try:
                gloo_group = dist.new_group(backend="gloo")
            except Exception as e:                          # no fallback then; the RCCL path is unaffected
                sys.stderr.write("bench.py rank %d: no gloo side channel (%s)\n" % (rank, str(e)[:200]))

        def make_sim(transport=None):
            sim = DomainDecomposedLBM(gN, D, nu, fp16c=fp16c, kernel=kern, device=local_rank, transport=transport)
            ox, oy, oz = sim.global_offset
            fl, u, rho = channel_state(sim.lNx, sim.lNy, sim.lNz, ox, oy, oz, *gN, buildings=args.buildings)
            sim.set_fields(fl, u, rho)
            if args.coriolis:
                sim.backend.set_coriolis(*coriolis_omega())
            return sim, fl
        sim = None
        try:
            sim, fl = make_sim()        # builds the RCCL connections to the neighbours, then allocates the lattice
            sim.initialize()
            ok = torch.ones(1)
        except Exception as e:      # RCCL p2p refused on this node: say so and fall back to host-staged halos rather than report nothing
            sys.stderr.write("bench.py rank %d: halo exchange over RCCL failed (%s); falling back to host-staged gloo\n" % (rank, str(e)[:300]))
            ok = torch.zeros(1)
        if gloo_group is None and ok.item() == 0:
            raise SystemExit("bench.py: halo exchange over RCCL failed and no fallback channel exists")
        if gloo_group is not None:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=gloo_group)
            if ok.item() == 0:
                from latticeurbanwind_amd.distributed import DomainLayout, HostStagedTransport
                if sim is not None: sim.backend.close()
                sim, fl = make_sim(HostStagedTransport(DomainLayout(gN, D, rank), group=gloo_group))
                sim.initialize()
                exchange_note = "host-staged gloo (RCCL point-to-point failed on this node)"
        sim.run(args.warmup)
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms = sim.run(args.steps, timed=True)
        torch.cuda.synchronize(); dist.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_device is not None else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        cells = Nx * Ny * Nz * world

    if rank == 0:
        mlups = cells * args.steps / dt / 1e6
        bpl = BYTES_PER_LUP[args.dtype] + (16.0 if args.every_step_fields else 0.0) + ((7 * 2 * (2 if fp16c else 4) + 4) if args.thermal else 0.0)   # + gi read/write + T write
        per_gpu_cells = Nx * Ny * Nz
        if D != (1, 1, 1) and sim.overlap:      # the timed launch is the interior box; the shell runs beside it on the other stream
            b = sim.layout.interior_box(); per_gpu_cells = (b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4])
        achieved = per_gpu_cells * bpl / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None
        out = {
            "metric": "MLUPS (D3Q19) at 1/2/4/8 MI355X; % of HBM roofline; u-field RMSE vs ref",
            "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if not fp16c else "fp16c-storage/f32-arithmetic", "data": "synthetic",
            "config": {"workload": "%s, log-law profile inflow on TYPE_E faces, solid ground, SRT+Smagorinsky, %s DDFs, rho/u written %s"
                       % (("%dx%dx%d D3Q19 channel%s" % (Nx, Ny, Nz, " with the configs[2] building array (solid fraction %.3f)" % float((fl == 1).mean()) if args.buildings else " (BASELINE configs[1])")) if world == 1 else
                          ("%dx%dx%d D3Q19 channel tile (8 GPUs: BASELINE configs[3]), %dx%dx%d cells per GPU" % (Nx * D[0], Ny * D[1], Nz * D[2], Nx, Ny, Nz)),
                          ("FP16C" if fp16c else "FP32") + (" + Coriolis force" if args.coriolis else "") + (" + thermal D3Q7 lattice" if args.thermal else ""), "every step" if args.every_step_fields else "by the last step only"),
                       "global_lattice": [Nx * D[0], Ny * D[1], Nz * D[2]], "n_gpu": list(D), "halo_exchange": (None if D == (1, 1, 1) else (exchange_note or ("gloo + host staging (--share-device test aid)" if args.share_device is not None else "RCCL p2p")) + (", overlapped with the interior (x rows kept whole)" if sim.overlap else " after the whole-box kernel (x split)")), "kernel": args.kernel,
                       "bytes_per_lup": bpl},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4) if achieved else None, "traffic": None,
                         "kernel_ms": round(kernel_ms, 4) if kernel_ms else None,
                         "whole_job_frac": round(mlups * 1e6 * bpl / 1e9 / (HBM_PEAK_GBPS * world), 4),   # wall-clock MLUPS of all GPUs x B/LUP over N x peak
                         "note": "achieved = %g B/LUP x %d cells / mean stream_collide duration (HIP events on the launch stream%s)" % (bpl, per_gpu_cells, "; interior box of rank 0, its boundary shell and the halo exchange run concurrently" if (D != (1, 1, 1) and sim.overlap) else "")},
        }
        # HBM traffic of the dominant kernel from rocprofv3 PMC counters: collected in separate --pmc passes of this same
        # command (tools/profile_bench.sh), corrected as MI355X_MICROARCH.md prescribes (read requests are 128 B), and
        # committed under profiles/; reported here only when that profile is of this workload
        prof = os.path.join(ROOT, "profiles", "r01g_pair_fp16c_512_summary.json" if (fp16c and args.kernel == "auto") else ("r01d_scalar_fp16c_512_summary.json" if fp16c else "r01g_scalar_f32_512_summary.json"))
        if (Nx, Ny, Nz) == (512, 512, 512) and args.kernel in ("auto", "scalar") and not args.every_step_fields and os.path.exists(prof):
            pr = json.load(open(prof))
            out["roofline"]["traffic"] = round(pr["hbm_traffic_bytes_per_launch"])
            out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(prof) + " (TCC_EA0_RDREQ x 128 B + WRITE_SIZE x 1024, per launch)"
        if world == 1 and not args.no_cpu_baseline and not args.force_distributed:
            out["cpu_baseline"] = cpu_baseline()
            try:
                out["parity"] = reference_parity()
            except Exception as e:      # never let the side measurement break the benchmark line
                out["parity"] = {"error": str(e)[:200]}
        sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(out)); sys.stdout.flush()
        os.dup2(2, 1)
    if world > 1 or args.force_distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
