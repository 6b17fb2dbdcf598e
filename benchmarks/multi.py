"""bench.py, N > 1: one process per GPU over RCCL (the driver's SCALE runs) -- the self-check against the CPU oracle through the real transport, the timed
tile in both cuts, the secondary blocks, the one-process multi-domain host as child processes, and the line that survives a failing block."""
import json
import os
import sys
import time

import numpy as np

from .common import BYTES_PER_LUP, HBM_PEAK_GBPS, NU, ROOT, coriolis_omega, device_context, fill_channel, tile_forcing
# ======================================================================== N > 1
PARITY_STEPS = 8                 # both time parities, eight exchanges per split axis
PROBE_STEPS = 20                 # real steps per schedule of the start-up probe (DomainDecomposedLBM.choose_schedule)
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
            from .line import emit as emit_line
            emit_line(self.fd, line if line is not None else self.line)       # short line on stdout, the full record in its file

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
                from .line import emit as emit_line
                sys.stdout.flush()
                emit_line(saved_stdout, {"metric": METRIC, "value": None, "unit": "MLUPS", "n_gpus": world, "per_rank": [],
                    "error": "decomposed run differs from the oracle: nothing was timed", "parity": parity})
            dist.barrier(); dist.destroy_process_group()
            raise SystemExit(3)

    def run_tile(D):
        """the tile cut as n_gpu = D: returns this rank's timing block; every failure (RCCL p2p included) propagates"""
        Bx, By, Bz = args.size or (512, 512, 512)
        gN = (Bx * D[0], By * D[1], Bz * D[2]) if args.size else tile_lattice(world)
        if any(g % d for g, d in zip(gN, D)):
            raise SystemExit("bench.py: lattice %s is not divisible by n_gpu %s" % (gN, D))
        kw = dict(buffer_nudging=nud, top_sponge=spg) if urban else {}
        # RCCL connections to the neighbours first, then the lattice (FP16C: --arith, native by default; the self-check above runs the bit-exact kernels)
        sim = DomainDecomposedLBM(gN, D, NU, fp16c=fp16c, kernel=kern, device=local_rank, native_arith=fp16c and args.arith != "exact", **kw)
        try:
            ox, oy, oz = sim.global_offset
            lb = sim.backend.lbm
            fill_channel(lb.flags.data, lb.u.data, lb.rho.data, sim.lNx, sim.lNy, sim.lNz, ox, oy, oz, *gN, buildings=urban)
            if args.coriolis:
                sim.backend.set_coriolis(*coriolis_omega())
            sim.initialize()
            # which step schedule this node wants depends on its wire: real steps under each, the slowest rank decides (x-split cuts only)
            probe = None
            if D[0] > 1 and os.environ.get("LUW_SCHEDULE_PROBE", "1") != "0":
                def slowest(v):
                    t = torch.tensor(v, dtype=torch.float64, device="cpu" if shared else "cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    return t.tolist()
                probe = sim.choose_schedule(steps=PROBE_STEPS, reduce_max=slowest)
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
            return {"D": D, "gN": gN, "dt": float(tmax.item()), "per_rank": per_rank, "overlap": sim.overlap, "one_phase": bool(sim.one_phase), "probe": probe,
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
            **({"arith": "exact" if args.arith == "exact" else "native"} if fp16c else {}),
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
                       "rccl_version": rccl, "ranks_in_communicator": dist.get_world_size(), "schedule_probe": res.get("probe")},
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
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--group-host-child", label, "--devices", ",".join(str(d) for d in devices), "--n-gpu",
            *(str(d) for d in D),
               "--global-lattice", *(str(g) for g in gN), "--dtype", args.dtype, "--arith", args.arith, "--kernel", args.kernel, "--steps",
                   str(min(args.steps, 60)), "--warmup",
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
    g = luw.LBMGroup(*gN, *D, NU, fp16c=fp16c, devices=devices, kernel=kern, global_arrays=False, native_arith=fp16c and args.arith != "exact", **kw)
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
