#!/usr/bin/env python3
"""bench.py -- MLUPS of the D3Q19 collide-stream hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its own ranks, benchmarks/launch.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N = 1 (the driver's BENCH line): BASELINE.json configs[2] ("C3", the largest single-GPU configuration): 1024x1024x256 D3Q19,
voxelised building cluster (SURVEY 8d closed-form box array), bounce-back, Smagorinsky LES, FP32 DDFs, z=0 plane solid, the
other five outer faces TYPE_E with a log-law inflow profile, interior initialised with the same profile, rho=1.  After the
headline a handful of `secondary` blocks are measured, each in a fresh child process (DEFAULT_BLOCKS: configs[1], the north
star's 1024^3 grid, C3 with FP16C DDFs / Coriolis / the thermal lattice, the undivided urban tile and one rank of the literal
n_gpu=[4,2,1] cut of configs[3] / configs[4]); --all-blocks adds the rest with their exact-arithmetic and peer-loopback twins,
--no-secondary skips them, --workload picks another headline.
N > 1 (the driver's SCALE runs): the urban tile of BASELINE configs[3] weak-scaled at 512^3 cells per GPU (8 GPUs:
2048x1024x512) -- building array, buffer nudging and top sponge with the deck defaults (160 m / 300 s, 200 m / 120 s), cut
as the deck's literal n_gpu (8 GPUs: [4,2,1], what BASELINE.json names; the x-whole cut [1,4,2] of the same tile is a secondary
block; --n-gpu picks another grid), one-cell halos over RCCL.  There is no fallback transport: if RCCL point-to-point fails
the run exits non-zero.
A "step" is one stream_collide pass over the whole lattice; rho/u are written by the last step only (153 B/LUP mode, see
DESIGN.md); data is synthetic and resident in HBM before the timed region.
Rank 0 prints ONE JSON line of at most 4 KB (benchmarks/line.py); the blocks in full go to gpurun_out/bench_secondary.json
(LUW_BENCH_FULL_JSON names another path), which the line cites as `secondary_file`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchmarks.common import (REFERENCE_GPU_CONTEXT, BYTES_PER_LUP, BUILDING_TOP, CELL_M, DT_S, HBM_PEAK_GBPS, NU, ROOT, WORKLOADS, attach_traffic,
    channel_state, coriolis_omega,
    cpu_baseline,
    cpu_model, device_context, fill_channel, loglaw_profile, profile_key, reference_case_rmse, reference_parity, tile_forcing, usable_cores)   # noqa: F401
from benchmarks.launch import needs_launcher, self_launch
from benchmarks.line import emit as emit_line
from benchmarks.multi import GROUP_HOST_VARIANTS, group_host_child, run_distributed   # noqa: F401



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
    attach_traffic(roof, profile_key(dtype, size, buildings, coriolis, thermal, every_step, urban, native and fp16c), kernel_name, placement["rows_per_xcd"])
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


# FP16C blocks that --all-blocks also measures with the bit-exact kernels (`--arith exact`: the forced / thermal / zone kernels the exact arithmetic costs
# most, and the plain one as the reference point)
EXACT_TWINS = ("c3_fp16c", "c3_fp16c_coriolis", "c3_fp16c_thermal", "tile512_urban_fp16c_coriolis", "c5_rank_4x2x1_fp16c_coriolis")


def run_single_block(luw, capi, device, key, native=False):
    wl, dt_, cor, th, urban = SINGLE_BLOCKS[key]
    sz, bld, _ = WORKLOADS[wl]
    # >= 200 timed after >= 20 warm-up steps
    r = run_single(luw, capi.KERNEL_AUTO, device, sz, dt_, bld, SECONDARY_STEPS, SECONDARY_WARMUP, coriolis=cor, thermal=th, urban=urban, native=native)
    r["workload"] = (
        "512^3 urban tile, undivided: the N = 1 point of the N > 1 lines (BASELINE configs[3]%s per GPU)" % (" / configs[4]" if cor else "")) if urban \
        else describe(wl, sz, bld, dt_, cor, th, False)
    return r


RANK_SHAPE_BLOCKS = {"c4_rank_4x2x1_f32": dict(fp16c=False, coriolis=False, D=(4, 2, 1), rank=0),       # the cut `--gpus 8` times: BASELINE's literal n_gpu
                     # the same tile with x kept whole (2048x258x258 per rank)
                     "c4_rank_1x4x2_f32": dict(fp16c=False, coriolis=False, D=(1, 4, 2), rank=7),
                     "c5_rank_4x2x1_fp16c_coriolis": dict(fp16c=True, coriolis=True, D=(4, 2, 1), rank=0),
                     "c5_rank_1x4x2_fp16c_coriolis": dict(fp16c=True, coriolis=True, D=(1, 4, 2), rank=7)}


def rank_shape_lattice(D):
    """the lattice a rank-shape block cuts: ONE tile whatever the cut (8 ranks: 2048x1024x512, so [1,4,2] ranks are 2048x258x258 and [4,2,1] ranks
    514x514x512 with their halos) -- tests/test_bench_line.py holds the blocks to the shapes tests/test_gpu_bench_workloads.py checks against the oracle"""
    from latticeurbanwind_amd.layout import tile_lattice
    return tile_lattice(D[0] * D[1] * D[2])


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
    gN = rank_shape_lattice(D)
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
            "workload": "rank %d of the %dx%dx%d urban tile as n_gpu=%s" % (rank, *gN, list(D)) if world > 1
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


# what the default N = 1 line measures after the headline (one fresh child process each); --all-blocks: every block, with its twins
DEFAULT_BLOCKS = ("c2_f32", "c3_fp16c", "c3_fp16c_coriolis", "c3_fp16c_thermal", "cube1024_f32", "cube1024_fp16c", "tile512_urban_fp16c_coriolis",
    "c4_rank_4x2x1_f32", "c5_rank_4x2x1_fp16c_coriolis")


DEFAULT_EXACT_TWINS = ("tile512_urban_fp16c_coriolis",)


def child_block(flag, key, local_rank, *extra):
    """ONE secondary block in a fresh child process (its JSON is the child's last stdout line); a failure is an `error` entry, never an exception"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), flag, key, *extra, "--steps", str(SECONDARY_STEPS), "--warmup", str(SECONDARY_WARMUP)],
            capture_output=True, text=True, timeout=600, env=dict(os.environ, LOCAL_RANK=str(local_rank)))
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": ("exit %d: " % r.returncode) + r.stderr[-300:]}
    except Exception as e:
        return {"error": str(e)[:300]}


def block_note(b):
    if "error" in b:
        return "error: " + str(b["error"])[:200]
    return "%.4f ms/step, %s of the HBM roofline" % (b["ms_per_step"], (b.get("roofline") or {}).get("frac"))


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
        help="domain grid Dx Dy Dz (default: the deck's literal cut, 4 2 1 on 8 GPUs; the x-whole cut 1 4 2 of the same tile is a secondary block)")
    ap.add_argument("--dtype", choices=["f32", "fp16c"], default="f32")
    ap.add_argument("--kernel", choices=["auto", "scalar", "pair"], default="auto")
    ap.add_argument("--arith", choices=["native", "exact"], default="native",
        help="FP16C collision arithmetic: native = LUW_OPT_NATIVE_ARITH (default, as in the deck driver: v_rcp / v_sqrt, fused multiply-adds -- as close to "
            "the "
             "real reference's fields as the exact kernels are, tests/test_gpu_native_arith.py), exact = bit-equal to the CPU oracle.  FP32 DDFs: no effect")
    ap.add_argument("--buildings", action="store_true",
        help="add the configs[2] solid mask to a workload that has none (c3 and the N > 1 tile always carry it)")
    ap.add_argument("--no-buildings", action="store_true", help="N > 1: plain channel tile without the building array / nudging / sponge")
    ap.add_argument("--coriolis", action="store_true", help="Coriolis body force at 31.25 deg N (BASELINE configs[4]): every cell takes the forced path")
    ap.add_argument("--thermal", action="store_true", help="also run the thermal D3Q7 lattice (the shipped reference build always does): +7 DDF planes and T")
    ap.add_argument("--urban", action="store_true", help="N = 1: add the urban tile's buffer nudging + top sponge (deck defaults) to the workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="only the headline measurement (profiling runs)")
    ap.add_argument("--all-blocks", action="store_true",
        help="N = 1: every secondary block (not only DEFAULT_BLOCKS), FP16C blocks also with exact arithmetic, "
        "rank shapes also with the peer-loopback transport: 25 child processes, about two minutes")
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

    # the bare `python3 bench.py --gpus N` (the driver's SCALE form): no launcher has set WORLD_SIZE, so this process starts its N ranks as a child
    # torch.distributed.run and exits with its code -- decided here, before torch or the HIP library are imported (benchmarks/launch.py)
    if needs_launcher(args.gpus, os.environ):
        raise SystemExit(self_launch(__file__, sys.argv[1:], args.gpus))

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
            blk = run_single_block(luw, capi, local_rank, args.secondary_block, native=args.arith != "exact")
        except Exception as e:
            blk = {"error": str(e)[:300]}
        sys.stdout.flush(); os.dup2(saved_stdout, 1)
        print(json.dumps(blk)); sys.stdout.flush()
        os.dup2(2, 1)
        return
    if args.rank_shape_block:
        blk = run_rank_shape(luw, torch, capi.KERNEL_AUTO, local_rank, steps=args.steps, warmup=args.warmup, native=args.arith != "exact",
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
            args.kernel, urban=args.urban, native=args.arith != "exact")
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
            sec = {}
            headline_key = next((k for k, v in SINGLE_BLOCKS.items() if v == (args.workload, args.dtype, args.coriolis, args.thermal, args.urban)), None)
            flag_of = lambda k: "--secondary-block" if k in SINGLE_BLOCKS else "--rank-shape-block"
            for key in (list(SINGLE_BLOCKS) + list(RANK_SHAPE_BLOCKS)) if args.all_blocks else DEFAULT_BLOCKS:
                if key == headline_key and not args.size:
                    continue
                sec[key] = child_block(flag_of(key), key, local_rank)
                sys.stderr.write("bench.py: secondary block %s: %s\n" % (key, block_note(sec[key]))); sys.stderr.flush()
                if "error" in sec[key]:
                    continue
                if not args.all_blocks:
                    # (the default line still shows what the FP16C arithmetic choice is worth on configs[4]'s physics: the exact twin of the urban tile)
                    if key in DEFAULT_EXACT_TWINS:
                        ex = child_block(flag_of(key), key, local_rank, "--arith", "exact")
                        sec[key]["exact"] = {k: ex[k] for k in ("ms_per_step", "arith", "error") if k in ex}
                        if "roofline" in ex: sec[key]["exact"]["roofline"] = {"frac": ex["roofline"]["frac"]}
                    continue
                # FP16C blocks: the same block once more with the bit-exact kernels, again in a fresh process; the block's own numbers are the default
                # arithmetic's ("arith": "native"), the twin sits under "exact"
                if key in EXACT_TWINS:
                    nat = child_block(flag_of(key), key, local_rank, "--arith", "exact")
                    sec[key]["exact"] = {k: nat[k] for k in ("value", "ms_per_step", "arith", "kernel_ms", "shell_ms", "exchange_ms", "error") if k in nat}
                    if "roofline" in nat:
                        sec[key]["exact"]["roofline"] = {k: nat["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "kernel_frac")
                            if k in nat["roofline"]}
                # rank shapes: the same rank once more with its faces written in place (the one-process host's peer stores) instead of RCCL self send / receive
                if key in RANK_SHAPE_BLOCKS:
                    tw = child_block(flag_of(key), key, local_rank, "--rank-transport", "peer-loopback")
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
            out["reference_context"] = REFERENCE_GPU_CONTEXT      # (full record only: benchmarks/line.py cites it as two numbers under cpu_baseline)
            try:
                out["parity"] = reference_parity()
            except Exception as e:      # never let the side measurement break the benchmark line
                out["parity"] = {"error": str(e)[:200]}
        sys.stdout.flush()
        emit_line(saved_stdout, out)            # the short line on stdout, everything above in gpurun_out/bench_secondary.json
        import torch.distributed as dist
        if dist.is_initialized():               # the one-rank world of the rank-shape blocks
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- N > 1: one process per GPU, halos over RCCL
    run_distributed(args, torch, luw, capi, kern, rank, world, local_rank, fp16c, METRIC, saved_stdout)



if __name__ == "__main__":
    main()
