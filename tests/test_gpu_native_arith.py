"""LUW_OPT_NATIVE_ARITH: the FP16C kernels' collision in the hardware's own arithmetic (one v_rcp_f32 for the divisions by the density, v_sqrt_f32 / v_rcp_f32
in the Smagorinsky rate, sums in trees, free contraction -- csrc/luw_device.hpp, collide_cell_pk_native).  Not bit-equal to the oracle by design; the gates:

  * ONE step from identical DDFs: u within 1e-8 RMSE of the oracle, rho bit-equal almost everywhere, at most 0.2 % of the stored FP16C codes different and
    none by more than one code unit -- the arithmetic itself is right to the last bits;
  * K = 8 / K = 64 on a deliberately noisy LES state: the differences of single stored codes (2^-12 relative each) grow like the flow lets them -- the same
    growth the REAL reference shows against the oracle (tests/golden/ref_shipped_*: 0.3-0.9e-6 at K = 8, 0.4-2.6e-5 at K = 64); recorded values x 2;
  * against the real reference's own fields: the same ceilings the bit-exact path has, and within twice the recorded values;
  * mass: the drift of the mean density over 1000 steps in a periodic box, worst of three seeds, no larger than twice the exact kernels' (+ 2e-8).
The bit-exact kernels stay the default and the anchor of every other test."""
import os

import numpy as np
import pytest

from helpers import synthetic_state, thermal_state, check_gate, rmse_u

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NUD = dict(n_cells=5, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1)
SPG = dict(n_cells=4, inv_tau=0.02)
COR = (0.0, 3e-5, 4e-5)


def pair(luw, Nx, Ny, Nz, st, forces, alpha=None, T=None):
    from oracle import oracle
    kw = dict(buffer_nudging=NUD, top_sponge=SPG) if "zones" in forces else {}
    g = luw.LBM(Nx, Ny, Nz, 2e-5, fp16c=True, native_arith=True, alpha=alpha, **kw)
    o = oracle.OracleLBM(Nx, Ny, Nz, 2e-5, fp16c=True, alpha=alpha)
    g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    if T is not None:
        g.T.data[:] = T; o.T[:] = T
    if "coriolis" in forces:
        g.set_coriolis(*COR); o.set_coriolis(*COR)
    if "zones" in forces:
        o.set_buffer_nudging(NUD["n_cells"], NUD["inv_tau"], NUD["downstream_face"], NUD["nudge_vertical"]); o.set_top_sponge(SPG["n_cells"], SPG["inv_tau"])
    return g, o


def code_distance(a, b):
    """FP16C codes as signed integers on the format's own grid (sign-magnitude -> two's complement): |difference| in code units"""
    mag = lambda c: (c.astype(np.int64) & 0x7FFF) * np.where(c & 0x8000, -1, 1)
    return np.abs(mag(np.asarray(a)) - mag(np.asarray(b)))


@pytest.mark.parametrize("forces", ["none", "coriolis", "zones+coriolis"])
@pytest.mark.parametrize("size", [(648, 28, 26), (130, 9, 7), (48, 20, 12)])      # pair kernel (even and ragged rows), one-cell kernel
def test_native_arithmetic_against_the_oracle(luw, size, forces):
    Nx, Ny, Nz = size
    st = synthetic_state(Nx, Ny, Nz, seed=21, shell="luw")
    g, o = pair(luw, Nx, Ny, Nz, st, forces)
    fluid = (st[0] & 3) == 0
    key = "native:%dx%dx%d:%s" % (Nx, Ny, Nz, forces)
    done = 0
    for K, ceiling in ((1, 1e-8), (8, 1e-5), (64, 5e-5)):
        g.run(K - done); o.run(K - done); done = K
        g.u.read_from_device(); g.rho.read_from_device()
        e = check_gate("%s:K%d" % (key, K), rmse_u(g.u.data, o.u, fluid), ceiling, "native vs oracle u RMSE at K=%d" % K)
        if K == 1:
            d = code_distance(g.download_fi(), o.fi)
            assert (d > 0).mean() < 2e-3 and d.max() <= 1, "one step: %.4f %% of the codes differ, by up to %d units" % (100 * (d > 0).mean(), d.max())
            assert np.abs(g.rho.data - o.rho).max() <= 2.4e-7 and np.abs(g.u.data - o.u).max() < 1e-7
        assert np.isfinite(g.u.data).all() and e >= 0.0
    g.close()


@pytest.mark.parametrize("size", [(514, 5, 6), (40, 12, 6)])
def test_native_arithmetic_with_the_thermal_lattice(luw, size):
    # T is advected with the velocity before the force shift: one step keeps T within a float ulp of the oracle's, the thermal codes within one unit
    Nx, Ny, Nz = size
    st = synthetic_state(Nx, Ny, Nz, seed=4, shell="luw")
    fl, T = thermal_state(st[0], size)
    fl = fl.copy(); fl[(fl & 3) == 2] |= 4
    g, o = pair(luw, Nx, Ny, Nz, (fl, st[1], st[2]), "coriolis", alpha=2e-5, T=T)
    g.run(1); o.run(1)
    g.T.read_from_device(); g.u.read_from_device()
    assert np.abs(g.T.data - o.T).max() < 5e-7
    d = code_distance(g.download_gi(), o.gi)
    assert d.max() <= 1 and (d > 0).mean() < 5e-3
    g.run(15); o.run(15); g.u.read_from_device(); g.T.read_from_device()
    check_gate("native:thermal:%dx%dx%d:K16" % size, rmse_u(g.u.data, o.u, (fl & 3) == 0), 2e-5, "native vs oracle u RMSE at K=16")
    assert np.abs(g.T.data - o.T).max() < 1e-3
    g.close()


@pytest.mark.parametrize("case", ["CaseA", "CaseL"])
def test_native_arithmetic_against_the_real_reference(luw, case):
    # the reference's shipped build (FP16C, -cl-mad-enable, native division) against this path in ITS native arithmetic: same ceilings as the exact path
    from oracle import setup_profile
    from test_oracle_vs_reference import run_and_compare
    gold = np.load(os.path.join(GOLD, "ref_shipped_%s.npz" % case))
    s = setup_profile.setup_profile_case(os.path.join(GOLD, "refcases", case, "conf.luwpf"), solid_mask=gold["solid"])
    nud = dict(n_cells=s["buffer_N"], inv_tau=float(s["buffer_inv_tau"]), downstream_face=s["buffer_face"], nudge_vertical=s["buffer_nudge_vertical"]) if s[
        "buffer_active"] else None
    spg = dict(n_cells=s["sponge_N"], inv_tau=float(s["sponge_inv_tau"])) if s["sponge_active"] else None
    g = luw.LBM(s["Nx"], s["Ny"], s["Nz"], float(s["nu"]), fp16c=True, buffer_nudging=nud, top_sponge=spg, native_arith=True)
    g.flags.data[:] = s["flags"]; g.u.data[:] = s["u"]; g.rho.data[:] = s["rho"]

    def dev_u(l):
        l.u.read_from_device(); return l.u.data

    def dev_rho(l):
        l.rho.read_from_device(); return l.rho.data
    run_and_compare(g, gold, s, "native:ref_shipped_%s" % case, 2e-6, 6e-5, dev_u, dev_rho)
    g.close()


def test_native_arithmetic_conserves_mass_like_the_exact_kernels(luw):
    # periodic box, no boundaries: the mean density moves by the rounding of the stored codes alone -- a random walk of either sign, 0.2-6e-8 after 1000 steps
    # for both arithmetics (tools/native_mass_drift.py, profiles/r05_native_raw_codec.txt).  Three seeds, worst case each: a bias in the native formulas (a
    # coefficient that does not cancel) would show as a drift of one sign growing with the step count.
    Nx, Ny, Nz = 256, 64, 64
    worst = {}
    for nat in (False, True):
        drift = []
        for seed in (5, 6, 7):
            st = synthetic_state(Nx, Ny, Nz, seed=seed, solids=False, shell=None)
            g = luw.LBM(Nx, Ny, Nz, 1e-4, fp16c=True, native_arith=nat)
            g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
            g.run(1); g.rho.read_from_device(); m0 = float(g.rho.data.astype(np.float64).mean())
            g.run(1000); g.rho.read_from_device(); drift.append(float(g.rho.data.astype(np.float64).mean()) - m0)
            g.close()
        worst[nat] = max(abs(d) for d in drift)
    assert worst[True] <= 2.0 * worst[False] + 2e-8 and worst[True] < 1.5e-7, worst


def test_native_mass_drift_is_a_random_walk_not_a_bias(luw):
    """what justifies the bound above: a bias of the native formulas -- a folded coefficient that does not cancel, a reciprocal that is always low
    -- moves the mean density by the SAME signed amount every step; rounding of the stored codes moves it by increments of either sign.  Per-step increments
    of the mean density over 600 steps, three seeds, both arithmetics: the mean increment of the native kernels must lie within four standard errors of zero
    or of the exact kernels' (whose encode is round-to-nearest-even: unbiased), and the walk must have ended where sqrt(K) growth puts it, not K growth."""
    Nx, Ny, Nz, K = 256, 64, 64, 600
    stats = {}
    for nat in (False, True):
        for seed in (5, 6, 7):
            st = synthetic_state(Nx, Ny, Nz, seed=seed, solids=False, shell=None)
            g = luw.LBM(Nx, Ny, Nz, 1e-4, fp16c=True, native_arith=nat)
            g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
            g.run(1)
            m = []
            for _ in range(K + 1):
                g.rho.read_from_device(); m.append(float(g.rho.data.astype(np.float64).mean())); g.run(1)
            g.close()
            d = np.diff(np.array(m))
            stats[(nat, seed)] = (d.mean(), d.std(ddof=1) / np.sqrt(len(d)), m[-1] - m[0], d.std(ddof=1))
    for seed in (5, 6, 7):
        (me, se_e, tot_e, sd_e), (mn, se_n, tot_n, sd_n) = stats[(False, seed)], stats[(True, seed)]
        print("seed %d: mean increment exact %+.2e +- %.1e, native %+.2e +- %.1e; total over %d steps %+.2e / %+.2e; sigma %.2e / %.2e" % (seed, me, se_e, mn,
            se_n, K, tot_e, tot_n, sd_e, sd_n))
        assert abs(mn) <= 4.0 * se_n or abs(mn - me) <= 4.0 * np.hypot(se_n, se_e), (seed, mn, se_n, me, se_e)
        assert abs(tot_n) <= 6.0 * sd_n * np.sqrt(K), (seed, tot_n, sd_n)          # a walk of K independent increments; a bias b would add b K
    # the three seeds do not all drift the same way by more than their own noise
    means = [stats[(True, s_)][0] for s_ in (5, 6, 7)]; ses = [stats[(True, s_)][1] for s_ in (5, 6, 7)]
    assert not (all(m_ > 3 * e_ for m_, e_ in zip(means, ses)) or all(m_ < -3 * e_ for m_, e_ in zip(means, ses))), (means, ses)


@pytest.mark.parametrize("forces", ["none", "zones+coriolis"])
def test_native_arithmetic_gives_a_cell_the_same_values_in_either_kernel(luw, forces):
    # a decomposed run takes the one-cell kernel where rows are narrow or unaligned and the pair kernel elsewhere: both run collide_cell_pk_native with the
    # populations scaled by 2^-112 from decode to encode, so a native run does not depend on how the lattice is cut -- bit for bit, like the exact kernels
    from latticeurbanwind_amd import capi
    Nx, Ny, Nz = 260, 14, 10
    st = synthetic_state(Nx, Ny, Nz, seed=9, shell="luw")
    # a region of fluid at rest behind a wall that starts at an odd x (lanes of the pair kernel that hold one solid and one fluid cell): u = 0 there must come
    # out with the same SIGN from both kernels (files are compared byte for byte: tests/test_gpu_driver.py::test_pair_and_scalar_kernels_write_identical_files)
    fl3, u4 = st[0].reshape(Nz, Ny, Nx), st[1].reshape(3, Nz, Ny, Nx)
    fl3[1:-1, 1:-1, 101:104] = 1; fl3[1:-1, 1:-1, 131:134] = 1; fl3[1:-1, 1:-1, 104:131] = 0
    u4[:, :, :, 100:135] = 0.0; st[2].reshape(Nz, Ny, Nx)[:, :, 100:135] = 1.0
    out = []
    for kern in (capi.KERNEL_SCALAR, capi.KERNEL_PAIR):
        kw = dict(buffer_nudging=NUD, top_sponge=SPG) if "zones" in forces else {}
        g = luw.LBM(Nx, Ny, Nz, 2e-5, fp16c=True, native_arith=True, kernel=kern, **kw)
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
        if "coriolis" in forces:
            g.set_coriolis(*COR)
        g.run(0); g.run(3); g.run(4)
        g.u.read_from_device(); g.rho.read_from_device()
        fi = np.asarray(g.download_fi()).copy(); fi[fi == 0x8000] = 0
        out.append((fi, g.u.data.copy(), g.rho.data.copy()))
        g.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32)) and np.array_equal(out[0][2],
        out[1][2])


@pytest.mark.parametrize("size,kernel", [((260, 14, 10), "pair"), ((48, 20, 12), "scalar")])
def test_native_sampled_steps_are_native_too(luw, size, kernel):
    # a native run is native from its first step to its last: the sampled steps carry the Welford update in the native kernels' epilogue, and that equals
    # the separate path { run(1); stats_accumulate() } on the fields the native kernels wrote, bit for bit (signs of zero included); the fields after the
    # window equal those of a plain native run of as many steps
    from latticeurbanwind_amd import capi
    Nx, Ny, Nz = size
    st = synthetic_state(Nx, Ny, Nz, seed=31, shell="luw")
    kern = capi.KERNEL_PAIR if kernel == "pair" else capi.KERNEL_SCALAR
    def make():
        g = luw.LBM(Nx, Ny, Nz, 2e-5, fp16c=True, native_arith=True, kernel=kern, buffer_nudging=NUD, top_sponge=SPG)
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]; g.set_coriolis(*COR)
        return g
    a, b, c = make(), make(), make()
    for g in (a, b, c): g.run(3)
    a.stats_reset(); b.stats_reset()
    a.run_sampled(7, 2, 2)                                             # samples at steps 2, 4, 6 of the window
    for i in range(1, 8):
        b.run(1)
        if i >= 2 and (i - 2) % 2 == 0: b.stats_accumulate()
    c.run(7)
    da, db = a.stats_download(), b.stats_download()
    assert da["count"] == db["count"] == 3
    for k in ("avg_u", "avg_rho", "m2_u", "m2_v", "m2_w"):
        assert np.array_equal(np.asarray(da[k]).view(np.uint32), np.asarray(db[k]).view(np.uint32)), k
    for g in (a, c): g.u.read_from_device(); g.rho.read_from_device()
    assert np.array_equal(a.u.data.view(np.uint32), c.u.data.view(np.uint32)) and np.array_equal(a.rho.data, c.rho.data)
    fa = np.asarray(a.download_fi()).copy(); fc = np.asarray(c.download_fi()).copy(); fa[fa == 0x8000] = 0; fc[fc == 0x8000] = 0
    assert np.array_equal(fa, fc)
    for g in (a, b, c): g.close()


def test_native_arithmetic_is_ignored_for_fp32(luw):
    # FP32 DDFs: the option changes nothing (bit-equal to the oracle)
    from oracle import oracle
    Nx, Ny, Nz = 48, 20, 12
    st = synthetic_state(Nx, Ny, Nz, seed=2, shell="luw")
    g = luw.LBM(Nx, Ny, Nz, 2e-5, native_arith=True)
    o = oracle.OracleLBM(Nx, Ny, Nz, 2e-5)
    g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    g.run(6); o.run(6); g.u.read_from_device()
    assert np.array_equal(g.u.data, o.u) and np.array_equal(g.download_fi(), o.fi)
    g.close()


def test_driver_option_arith_native(luw, tmp_path):
    # --arith through the deck driver (native is its default for FP16C): files within the exact run's gates of the real reference, and different from the
    # exact run's files
    import glob, shutil, subprocess
    from vtkio import read_vtk
    drv = os.path.join(os.path.dirname(GOLD), "..", "latticeurbanwind_amd", "host", "luw_driver")
    subprocess.check_call(["make", "-C", os.path.dirname(drv), "-s"])
    gold = np.load(os.path.join(GOLD, "ref_shipped_CaseA.npz"))
    out = {}
    for mode in ("exact", "native"):
        proj = str(tmp_path / mode)
        shutil.copytree(os.path.join(GOLD, "refcases", "CaseA"), proj)
        r = subprocess.run([drv, os.path.join(proj, "conf.luwpf"), "--ddf", "fp16c", "--arith", mode], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert ("native (v_rcp" in r.stdout) == (mode == "native") and ("exact (bit-equal" in r.stdout) == (mode == "exact")
        out[mode] = read_vtk(glob.glob(os.path.join(proj, "RESULTS", "vtk", "*_raw_u-000000064.vtk"))[0])[1]["data"]
    fac = np.float32(5.0) / np.float32(0.1)
    fluid = ~gold["solid"]
    err = lambda a: float(np.sqrt(((((a - gold["u64"]) / fac)[fluid].astype(np.float64)) ** 2).sum(-1).mean()))
    assert not np.array_equal(out["exact"], out["native"])
    check_gate("driver:native:ref_shipped_CaseA:final", err(out["native"]), 6e-5)
    assert err(out["exact"]) < 6e-5
