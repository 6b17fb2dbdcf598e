"""End to end on the GPU: the C++ driver (`FluidX3D <deck>` replacement) reads a deck + profile + STL, runs the HIP core and
writes RESULTS/vtk/*; its files must equal what the CPU restatements produce for the same deck (set-up restatement +
oracle kernel + host Welford + the avg-VTK formulas of FX/setup.cpp:2513-2683), value for value."""
import glob
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle, setup_profile
from vtkio import read_vtk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")


@pytest.mark.parametrize("case,ddf", [("CaseA", "fp32"), ("CaseB", "fp16c")])
def test_driver_outputs_equal_cpu_restatement(luw, tmp_path, case, ddf):
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    proj = str(tmp_path / case)
    shutil.copytree(os.path.join(GOLD, "refcases", case), proj)
    deck = os.path.join(proj, "conf.luwpf")
    r = subprocess.run([DRIVER, deck, "--ddf", ddf], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    s = setup_profile.setup_profile_case(deck)
    Nx, Ny, Nz, Nzc = s["Nx"], s["Ny"], s["Nz"], s["Nz_core"]
    o = oracle.OracleLBM(Nx, Ny, Nz, s["nu"], fp16c=(ddf == "fp16c"))
    o.flags[:] = s["flags"]; o.u[:] = s["u"]; o.rho[:] = s["rho"]
    if s["buffer_active"]:
        o.set_buffer_nudging(s["buffer_N"], s["buffer_inv_tau"], s["buffer_face"], s["buffer_nudge_vertical"])
    if s["sponge_active"]:
        o.set_top_sponge(s["sponge_N"], s["sponge_inv_tau"])
    stats = oracle.OracleStats(o.N)
    fu, fr = s["si_u_factor"], s["si_rho_factor"]
    vt = os.path.join(proj, "RESULTS", "vtk")
    for t in range(1, 65):
        o.run(1)
        if t >= 61:                                            # purge_avg = 4 of run_nstep = 64
            stats.accumulate(o)
        if t % 8 == 0:                                         # unsteady_output = 8
            h, f = read_vtk(glob.glob(os.path.join(vt, "*_raw_u-%09d.vtk" % t))[0])
            mine = (o.u.reshape(3, Nz, Ny, Nx)[:, :Nzc] * fu).astype(np.float32).transpose(1, 2, 3, 0)
            assert h["dims"] == (Nx, Ny, Nzc) and np.array_equal(f["data"], mine), "u at t=%d" % t
    h, f = read_vtk(glob.glob(os.path.join(vt, "*_raw_rho-000000064.vtk"))[0])
    assert np.array_equal(f["data"][..., 0], (o.rho.reshape(Nz, Ny, Nx)[:Nzc] * fr).astype(np.float32))
    h, f = read_vtk(glob.glob(os.path.join(vt, "*_avg-000000064.vtk"))[0])
    pts = Nx * Ny * Nzc
    ua = stats.avg_u[:3 * pts].reshape(Nzc, Ny, Nx, 3)
    assert np.array_equal(f["u_avg"], (ua * fu).astype(np.float32))
    assert np.array_equal(f["rho_avg"][..., 0], (stats.avg_rho[:pts].reshape(Nzc, Ny, Nx) * fr).astype(np.float32))
    solid = (s["flags"][:pts].reshape(Nzc, Ny, Nx) & 1) != 0
    assert np.array_equal(f["fluid"][..., 0] == 0, solid)
    inv_n = np.float32(1.0) / np.float32(4)
    var = [np.maximum(m[:pts].reshape(Nzc, Ny, Nx) * inv_n, np.float32(0)) for m in (stats.m2_u, stats.m2_v, stats.m2_w)]
    var_sum = (var[0] + var[1]) + var[2]
    tke = np.where(solid, np.float32(0), np.float32(0.5) * var_sum) * (fu * fu)
    assert np.array_equal(f["tke"][..., 0], tke.astype(np.float32))
    umag = np.sqrt((ua[..., 0] * ua[..., 0] + ua[..., 1] * ua[..., 1]) + ua[..., 2] * ua[..., 2])
    with np.errstate(divide="ignore", invalid="ignore"):
        ti = np.where((~solid) & (umag > 1e-9) & (var_sum > 0), np.sqrt(var_sum * (np.float32(1.0) / np.float32(3.0))) / umag, np.float32(0)).astype(np.float32)
    assert np.allclose(f["TI"][..., 0], ti, rtol=2e-7, atol=0)
    assert np.isfinite(f["TLS"]).all() and (f["TLS"][..., 0][solid] == 0).all() and f["TLS"].max() <= max(Nx, Ny, Nzc) * h["spacing"][0]
    # the same deck through the real reference: identical header grammar (FX/lbm.hpp:322-329)
    gold = np.load(os.path.join(GOLD, "ref_fp32_%s.npz" % case))
    assert tuple(gold["dims"]) == h["dims"] and np.allclose(gold["origin"], h["origin"]) and np.allclose(gold["spacing"], h["spacing"])
