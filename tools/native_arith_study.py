#!/usr/bin/env python3
"""Native-arithmetic FP16C kernels (LUW_OPT_NATIVE_ARITH) against the bit-exact ones: value differences after 1 / 8 / 64 steps against the CPU oracle, against
the fields of the real reference (tests/golden/ref_shipped_*), mass drift, and step times on the bench workloads.  GPU box; prints a report.
usage: tools/native_arith_study.py [--no-timing] [--steps N]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)


def rmse(ua, ub, mask):
    d = (ua.reshape(3, -1) - ub.reshape(3, -1))[:, mask].astype(np.float64)
    return float(np.sqrt((d ** 2).sum(0).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    import latticeurbanwind_amd as luw
    from oracle import oracle, setup_profile
    from helpers import synthetic_state
    luw.build(); luw.load()
    nud = dict(n_cells=5, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1); spg = dict(n_cells=4, inv_tau=0.02)
    print("== native vs oracle (exact kernels equal the oracle bit for bit), 648x28x26 FP16C, synthetic urban state")
    for name, cor, zones in (("force-free", None, False), ("coriolis", (0.0, 3e-5, 4e-5), False), ("zones+coriolis", (0.0, 3e-5, 4e-5), True)):
        Nx, Ny, Nz = 648, 28, 26
        st = synthetic_state(Nx, Ny, Nz, seed=21, shell="luw")
        kw = dict(buffer_nudging=nud, top_sponge=spg) if zones else {}
        g = luw.LBM(Nx, Ny, Nz, 2e-5, fp16c=True, native_arith=True, **kw)
        o = oracle.OracleLBM(Nx, Ny, Nz, 2e-5, fp16c=True)
        for l, d in ((g, True), (o, False)):
            (l.flags.data if d else l.flags)[:] = st[0]; (l.u.data if d else l.u)[:] = st[1]; (l.rho.data if d else l.rho)[:] = st[2]
        if cor: g.set_coriolis(*cor); o.set_coriolis(*cor)
        if zones: o.set_buffer_nudging(nud["n_cells"], nud["inv_tau"], nud["downstream_face"], nud["nudge_vertical"]); o.set_top_sponge(spg["n_cells"],
                spg["inv_tau"])
        fluid = (st[0] & 3) == 0
        done = 0
        for K in (1, 8, 64):
            g.run(K - done); o.run(K - done); done = K
            g.u.read_from_device(); g.rho.read_from_device()
            fi = np.asarray(g.download_fi()).astype(np.int64); fo = np.asarray(o.fi).astype(np.int64)
            mag = lambda c: (c & 0x7FFF) * np.where(c & 0x8000, -1, 1)
            dc = np.abs(mag(fi) - mag(fo))
            du = np.abs(g.u.data - o.u)
            print("  %-15s K=%-3d u-RMSE %.3e  max|du| %.3e  max|drho| %.3e  DDF codes differing %.4f %%  (by >1 code unit: %d of %d)" % (
                name, K, rmse(g.u.data, o.u, fluid), du.max(), np.abs(g.rho.data - o.rho).max(), 100.0 * (dc > 0).mean(), int((dc > 1).sum()), dc.size))
        g.close()
    print("== against the REAL reference (shipped FP16C build), 48x40x28, the one-cell kernel: u-RMSE at K = 8 / 64, oracle | exact HIP | native HIP")
    from test_oracle_vs_reference import make_oracle
    GOLD = os.path.join(ROOT, "tests", "golden")
    for case in ("CaseA", "CaseL"):
        gold = np.load(os.path.join(GOLD, "ref_shipped_%s.npz" % case))
        s = setup_profile.setup_profile_case(os.path.join(GOLD, "refcases", case, "conf.luwpf"), solid_mask=gold["solid"])
        n_ = dict(n_cells=s["buffer_N"], inv_tau=float(s["buffer_inv_tau"]), downstream_face=s["buffer_face"], nudge_vertical=s["buffer_nudge_vertical"]) if s[
            "buffer_active"] else None
        sp = dict(n_cells=s["sponge_N"], inv_tau=float(s["sponge_inv_tau"])) if s["sponge_active"] else None
        runs = {"oracle": make_oracle(s, True)}
        for nm, nat in (("exact", False), ("native", True)):
            g = luw.LBM(s["Nx"], s["Ny"], s["Nz"], float(s["nu"]), fp16c=True, buffer_nudging=n_, top_sponge=sp, native_arith=nat)
            g.flags.data[:] = s["flags"]; g.u.data[:] = s["u"]; g.rho.data[:] = s["rho"]
            runs[nm] = g
        Nx, Ny, Nz, Nzc = s["Nx"], s["Ny"], s["Nz"], s["Nz_core"]; fac = s["si_u_factor"]; fluid = ~gold["solid"]
        def vs_ref(u_now, t):
            mine = (u_now.reshape(3, Nz, Ny, Nx)[:, :Nzc] * fac).astype(np.float32).transpose(1, 2, 3, 0)
            d = ((mine - gold["u%d" % t]) / fac)[fluid].astype(np.float64)
            return float(np.sqrt((d ** 2).sum(-1).mean()))
        out = {}
        for nm, l in runs.items():
            l.run(8)
            if nm != "oracle": l.u.read_from_device()
            u8 = (l.u if nm == "oracle" else l.u.data).copy()
            l.run(56)
            if nm != "oracle": l.u.read_from_device()
            u64 = (l.u if nm == "oracle" else l.u.data).copy()
            out[nm] = (vs_ref(u8, 8), vs_ref(u64, 64), u8, u64)
        fl = (s["flags"] & 3) == 0
        print("  %s  K=8: %.3e | %.3e | %.3e    K=64: %.3e | %.3e | %.3e    native vs oracle: K=8 %.3e K=64 %.3e" % (case, out["oracle"][0], out["exact"][0],
            out["native"][0], out["oracle"][1], out["exact"][1], out["native"][1], rmse(out["native"][2], out["oracle"][2], fl),
                rmse(out["native"][3], out["oracle"][3], fl)))
        runs["exact"].close(); runs["native"].close()
    print("== mass drift, 256x64x64 periodic box at rest + noise, FP16C, 200 steps: sum(rho)/N - 1 at t=0 / exact / native")
    Nx, Ny, Nz = 256, 64, 64
    st = synthetic_state(Nx, Ny, Nz, seed=5, solids=False, shell=None)
    res = []
    for nat in (False, True):
        g = luw.LBM(Nx, Ny, Nz, 1e-4, fp16c=True, native_arith=nat)
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
        g.run(1); g.rho.read_from_device(); m0 = float(g.rho.data.astype(np.float64).mean())
        g.run(200); g.rho.read_from_device(); res.append((m0, float(g.rho.data.astype(np.float64).mean())))
        g.close()
    print("  t=1: %.9f / %.9f   t=201: exact %.9f (drift %.3e)  native %.9f (drift %.3e)" % (res[0][0], res[1][0], res[0][1], res[0][1] - res[0][0], res[1][1],
        res[1][1] - res[1][0]))
    if args.no_timing:
        return
    import bench
    from latticeurbanwind_amd import capi
    print("== step time (mean stream_collide duration over %d steps, fresh solver each, exact then native, twice)" % args.steps)
    for key in ("c3_fp16c", "c3_fp16c_coriolis", "c3_fp16c_thermal", "tile512_urban_fp16c_coriolis", "c2_fp16c"):
        wl, dt_, cor, th, urban = bench.SINGLE_BLOCKS[key]
        sz, bld, _ = bench.WORKLOADS[wl]
        row = []
        for rep in range(2):
            for nat in (False, True):
                r = bench.run_single(luw, capi.KERNEL_AUTO, 0, sz, dt_, bld, args.steps, 20, coriolis=cor, thermal=th, urban=urban, native=nat)
                row.append("%s %.3f ms (%.3f)" % ("native" if nat else "exact", r["roofline"]["kernel_ms"], r["roofline"]["frac"]))
        print("  %-30s %s" % (key, "   ".join(row))); sys.stdout.flush()


if __name__ == "__main__":
    main()
