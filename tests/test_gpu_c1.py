"""BASELINE configs[0] ("C1") literally, at the horizons SURVEY 8(d) names: the 128^3 profile-inflow box without solids -- z = 0 plane TYPE_S, the
other five outer faces TYPE_E, the 12-point profile of example_ProfileResearch_noDEM cubic-Hermite interpolated on a 0.1 m table (nearest
index), cell = 2 m, base_height = 0, wind along +x (deck angle 270), si_ref_u = 7.8 m/s at u_lbm = 0.1, interior initialised with the
same profile, rho = 1 -- stepped by the HIP path and by the CPU oracle from identical initial DDFs:

  * turbulent regime, nu = units.nu(1.48e-5) (tau 4 ulp above 1/2, LES carries the stability), K = 100 steps, FP32 and FP16C;
  * laminar regime, nu = 0.1 / 6, K = 1000 steps, FP32.

Bar: bit for bit (rho, u, all 19 DDF planes), which is far inside the north star's 1e-5 RMSE gate at these horizons; the oracle itself is
pinned to the real reference at K = 64 on the committed cases (tests/test_oracle_vs_reference.py).

And against the REAL reference at this size: the same configuration as a deck (tests/golden/refcases/CaseC1: 128^3 cells of 2 m, that profile, a ground
slab, K = 100) went through both builds of the reference on an MI355X (tools/reference_session_c1.sh); three orthogonal mid-planes of u at K = 100 and of
u_avg, with the fields' global minimum / maximum / mean, are committed as tests/golden/ref_{fp32,shipped}_C1_planes.npz.  test_c1_deck_against_the_real_
reference runs the deck driver on the deck and holds its planes to them."""
import glob
import math
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# examples/example_ProfileResearch_noDEM/wind_bc/profile.dat of the reference: data (height above ground in m, speed in m/s)
PROFILE = [(1.25, 2.847), (2.5, 3.042), (5, 3.2604), (7.5, 3.4086), (12.5, 3.7674), (25, 4.3602), (50, 5.109), (75, 5.694), (100, 6.162), (150, 6.9654),
    (200, 7.3944), (250, 7.838)]
N = 128
CELL_M = 2.0


def c1_state():
    """(flags, u, rho, nu_turbulent) of the C1 recipe, host arithmetic as oracle/setup_profile.py restates it (FX/setup.cpp:3731-3737,5777-5912)"""
    from oracle import setup_profile as sp
    f32 = np.float32
    zv = [f32(z) for z, _ in PROFILE]; uv = [f32(u) for _, u in PROFILE]
    si_ref_u = max(uv); lbm_ref_u = f32(0.10)
    units = sp.Units()
    units.set_m_kg_s_K(f32(N), lbm_ref_u, f32(1), f32(1), f32(N * CELL_M), si_ref_u, f32(1.225), f32(293.15))
    u_scale = lbm_ref_u / si_ref_u
    dz = f32(0.1); inv_dz = f32(1.0) / dz
    table_top = units.si_x(f32(N - 1))
    steps = int(math.ceil(float(table_top / dz)))
    table = []
    for i in range(steps + 1):
        v = sp.interpolate_profile_cubic(zv, uv, min(table_top, f32(i) * dz))
        table.append((v if v >= 0 else f32(0)) * u_scale)
    table = np.array(table, f32)
    spd = np.zeros(N, f32)
    for z in range(1, N):                                     # z = 0 is the ground plane itself (pos_z <= ground_z: 0)
        z_agl = units.si_x(f32(z))
        spd[z] = table[min(max(0, sp.lround(z_agl * inv_dz)), len(table) - 1)]
    flags = np.zeros((N, N, N), np.uint8)
    flags[:, :, 0] = flags[:, :, -1] = flags[:, 0, :] = flags[:, -1, :] = 2; flags[-1] = 2
    flags[0] = 1
    u = np.zeros((3, N, N, N), f32)
    u[0] = spd[:, None, None]; u[0][flags == 1] = 0
    return flags.ravel(), u.ravel(), np.ones(N ** 3, f32), float(units.nu(f32(1.48e-5))), float(spd.max())


@pytest.mark.parametrize("regime,fp16c,steps", [("turbulent", False, 100), ("turbulent", True, 100), ("laminar", False, 1000)])
def test_c1_literal(luw, regime, fp16c, steps):
    from oracle import oracle
    flags, u, rho, nu_t, umax = c1_state()
    assert abs(umax - 0.1 * 7.838 / 7.838) < 2e-3 and abs(nu_t - 1.48e-5 * (CELL_M * 0.1 / 7.838) / CELL_M ** 2) < 1e-9
    nu = nu_t if regime == "turbulent" else 0.1 / 6.0
    g = luw.LBM(N, N, N, nu, fp16c=fp16c); o = oracle.OracleLBM(N, N, N, nu, fp16c=fp16c)
    g.flags.data[:] = flags; g.u.data[:] = u; g.rho.data[:] = rho
    o.flags[:] = flags; o.u[:] = u; o.rho[:] = rho
    for chunk in (steps // 2, steps - steps // 2):             # two calls: both the "fields written by the last step" boundary and odd/even t
        g.run(chunk); o.run(chunk)
    g.u.read_from_device(); g.rho.read_from_device()
    fluid = (flags & 1) == 0
    d = (g.u.data.reshape(3, -1) - o.u.reshape(3, -1))[:, fluid].astype(np.float64)
    rmse = float(np.sqrt((d ** 2).sum(0).mean()))
    assert rmse == 0.0 and np.array_equal(g.u.data, o.u) and np.array_equal(g.rho.data, o.rho), "%s K=%d: u RMSE vs oracle %.3e" % (regime, steps, rmse)
    fi = np.asarray(g.download_fi()); ref = o.fi
    if fp16c:
        fi = np.where(fi == 0x8000, 0, fi); ref = np.where(ref == 0x8000, 0, ref)
    assert np.array_equal(fi, ref)
    uy = np.abs(g.u.data.reshape(3, -1)[1]).max()
    assert np.isfinite(g.u.data).all() and float(np.abs(g.u.data).max()) <= 0.57735027 + 1e-6
    if regime == "turbulent":
        assert uy > 0.0                                        # the flow left the pure profile (corner cells of the TYPE_E shell shed it)
    g.close()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")
SI_PER_LU = np.float32(7.838) / np.float32(0.1)           # si_ref_u = the profile's maximum at u_lbm = 0.1


def planes_of(a):
    nz, ny, nx = a.shape[:3]
    return {"xy": a[nz // 2], "xz": a[:, ny // 2], "yz": a[:, :, nx // 2]}


@pytest.mark.parametrize("ddf,arith,fixture,ceiling",
    [("fp32", "exact", "ref_fp32", 1e-5), ("fp16c", "native", "ref_shipped", 1e-4), ("fp16c", "exact", "ref_shipped", 1e-4),
    ("fp16c", "native", "ref_fp32", 1e-4)])
def test_c1_deck_against_the_real_reference(luw, tmp_path, ddf, arith, fixture, ceiling):
    """FP32 DDFs: inside the north star's 1e-5 with a factor of forty (2.5e-7 at K = 100).  FP16C DDFs: the shipped precision against the shipped build, native
    (the driver's default) and exact arithmetic, and the native run against the reference's FP32 build -- every such pair sits at 3-4e-5, which is where the
    reference's shipped build sits against its OWN FP32 build (3.2e-5, test_the_reference_at_c1_size_against_itself): recorded values x 2 (check_gate)."""
    from helpers import check_gate
    sys.path.insert(0, GOLD)
    from vtkio import read_vtk
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    proj = str(tmp_path / "CaseC1")
    shutil.copytree(os.path.join(GOLD, "refcases", "CaseC1"), proj)
    r = subprocess.run([DRIVER, os.path.join(proj, "conf.luwpf"), "--ddf", ddf, "--arith", arith], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    gold = np.load(os.path.join(GOLD, fixture + "_C1_planes.npz"))
    vt = os.path.join(proj, "RESULTS", "vtk")
    h, f = read_vtk(glob.glob(os.path.join(vt, "*_raw_u-000000100.vtk"))[0])
    ha, fa = read_vtk(glob.glob(os.path.join(vt, "*_avg-000000100.vtk"))[0])
    assert tuple(h["dims"]) == tuple(int(v) for v in gold["dims"]) == (128, 128, 128)
    assert int((fa["fluid"][..., 0] == 0).sum()) == int(gold["solid_count"])                 # the same voxels are solid
    for name, mine in (("u100", f["data"]), ("u_avg", fa["u_avg"])):
        sq, cells = 0.0, 0
        for pl, a in planes_of(mine).items():
            fluid = ~gold["solid_" + pl]
            d = ((a - gold["%s_%s" % (name, pl)]) / SI_PER_LU)[fluid].astype(np.float64)
            sq += float((d ** 2).sum()); cells += int(fluid.sum())
        rm = math.sqrt(sq / cells)
        check_gate("c1deck:%s:%s:%s:%s" % (ddf, arith, fixture, name), rm, ceiling, "%s RMSE over the three mid-planes, lattice units" % name)
        if ddf == "fp32":
            assert rm < 1e-6
        fl3 = fa["fluid"][..., 0] != 0
        st = np.array([mine[fl3].min(0), mine[fl3].max(0), mine[fl3].astype(np.float64).mean(0)])
        assert np.abs(st - gold[name + "_stats"]).max() / SI_PER_LU < (2e-6 if ddf == "fp32" else 2e-3), (name, st, gold[name + "_stats"])
