"""Pins the CPU oracle (kernel restatement + profile-mode set-up restatement) against fields produced by the REAL
reference solver run on an MI355X through OpenCL (tests/golden/ref_*.npz, see tests/golden/README.md).

Tolerances (lattice units, stated per the north star): the reference is compiled by the OpenCL driver with
-cl-mad-enable, so individual operations may be fused differently from the oracle's fixed evaluation; with FP32
DDFs the fields agree to ~1e-7 RMSE after 64 steps (gate 1e-6, an order below the 1e-5 acceptance gate); with
FP16C DDFs every differing last bit can flip an 11-bit mantissa rounding (2^-12 relative), giving ~3e-7 RMSE
after 8 steps and ~2.5e-5 after 64 steps of LES flow (gates 2e-6 / 1e-4)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle, setup_profile

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_case(case, ref_npz, fp16c, make_lbm):
    g = np.load(os.path.join(GOLD, ref_npz))
    s = setup_profile.setup_profile_case(os.path.join(GOLD, "refcases", case, "conf.luwpf"), solid_mask=g["solid"])
    lbm = make_lbm(s, fp16c)
    return g, s, lbm


def make_oracle(s, fp16c):
    o = oracle.OracleLBM(s["Nx"], s["Ny"], s["Nz"], s["nu"], fp16c=fp16c)
    o.flags[:] = s["flags"]; o.u[:] = s["u"]; o.rho[:] = s["rho"]
    if s["buffer_active"]:
        o.set_buffer_nudging(s["buffer_N"], s["buffer_inv_tau"], s["buffer_face"], s["buffer_nudge_vertical"])
    if s["sponge_active"]:
        o.set_top_sponge(s["sponge_N"], s["sponge_inv_tau"])
    return o


def compare(g, s, u_now, rho_now, t, gate_rmse, key=None, field=None):
    """u RMSE (lattice units, non-solid cells) against the reference's field `field` (default u<t>); key: the entry of tests/golden/observed_rmse.json that
    also bounds it at twice the recorded value (helpers.check_gate)"""
    from helpers import check_gate
    Nx, Ny, Nz, Nzc = s["Nx"], s["Ny"], s["Nz"], s["Nz_core"]
    fac = s["si_u_factor"]
    mine = (u_now.reshape(3, Nz, Ny, Nx)[:, :Nzc] * fac).astype(np.float32).transpose(1, 2, 3, 0)  # SI, like write_vtk
    ref = g[field or "u%d" % t]
    fluid = ~g["solid"]
    d = ((mine - ref) / fac)[fluid].astype(np.float64)
    rmse = float(np.sqrt((d ** 2).sum(-1).mean()))
    if key: check_gate(key, rmse, gate_rmse, "u RMSE at t=%d" % t)
    assert rmse < gate_rmse, "u RMSE %.3e at t=%d" % (rmse, t)
    if rho_now is not None:
        mr = (rho_now.reshape(Nz, Ny, Nx)[:Nzc] * s["si_rho_factor"]).astype(np.float32)
        dr = np.abs((mr - g["rho64"]) / s["si_rho_factor"])[fluid].max()
        assert dr < 100 * gate_rmse, "rho max diff %.3e" % dr
    return rmse


def test_host_setup_matches_reference_console():
    # host-stage goldens printed by the real binary (tests/golden/ref_fp32_CaseA.console.txt)
    g = np.load(os.path.join(GOLD, "ref_fp32_CaseA.npz"))
    s = setup_profile.setup_profile_case(os.path.join(GOLD, "refcases", "CaseA", "conf.luwpf"), solid_mask=g["solid"])
    txt = open(os.path.join(GOLD, "ref_fp32_CaseA.console.txt")).read()
    assert "Grid Resolution | 48,   40,   28 (nCell = 53760)" in txt and (s["Nx"], s["Ny"], s["Nz"]) == (48, 40, 28)
    assert "core Nz=24, ext=4, total Nz=28" in txt and s["Nz_core"] == 24
    assert "1 cell = 2000.000 mm, 1 s = 25 time steps" in txt and s["units"].t(1.0) == 25
    assert "Nbuf=4 cells" in txt and s["buffer_N"] == 4
    assert "inv_tau_lbmu=0.01333333, downstream_face_id=auto" in txt and abs(float(s["buffer_inv_tau"]) - 0.01333333) < 1e-8
    assert "Nsponge=4 cells" in txt and s["sponge_N"] == 4
    assert "inv_tau_lbmu=0.02000000, ref_mode=0" in txt and abs(float(s["sponge_inv_tau"]) - 0.02) < 1e-8
    assert "side_ref_cap_z=23" in txt and s["side_ref_z_cap"] == 23
    assert "scaled by 0.5000" in txt and float(s["scale_geom"]) == 0.5
    assert "AGL top=50.000 m, core_top=46.000 m, solver_top=54.000 m" in txt
    assert float(s["table_top"]) == 50.0 and float(s["core_top_si"]) == 46.0 and float(s["solver_top_si"]) == 54.0
    assert "Downstream BC   | +x" in txt and s["downstream_bc"] == "+x"
    assert "profile boundaries mapped: 6048 cells" in txt and s["mapped_bc"] == 6048
    assert "1.9000 to 5.0237" in txt and abs(float(s["prof_si"].max()) - 5.0237) < 5e-5 and float(s["prof_si"].min()) == np.float32(1.9)
    assert np.array_equal((s["flags"].reshape(28, 40, 48)[:24] & 1) != 0, g["solid"])


def test_voxeliser_restatement_counts():
    # the z-ray voxeliser (FX/kernel.cpp:2381-2471) restated with IEEE division marks the same NUMBER of cells as the
    # reference on the GPU (2616, console line "Voxelized cells ... solid = 2616"); the slab layer differs by one cell
    # in z because faces sit exactly on lattice planes and the GPU reciprocal rounds 1/g differently (see README).
    s = setup_profile.setup_profile_case(os.path.join(GOLD, "refcases", "CaseB", "conf.luwpf"))
    assert s["n_solid_vox"] == 2616
    assert s["terrain_solid_bc"] == 259 and s["mapped_bc"] == 5360
    txt = open(os.path.join(GOLD, "ref_fp32_CaseB.console.txt")).read()
    assert "solid = 2616" in txt and "mapped: 5360 cells" in txt and "-> solid: 259" in txt


def run_and_compare(lbm, g, s, fixture, ceil8, ceil64, get_u=lambda l: l.u, get_rho=lambda l: l.rho):
    """8 steps, 52 more, then the last four one by one (u_avg of the fixtures = the reference's mean over its last 4 steps, FX/setup.cpp:4441-4488)"""
    lbm.run(8)
    r8 = compare(g, s, get_u(lbm), None, 8, ceil8, key=fixture + ":u8")
    lbm.run(52)
    mean = None
    for k in range(4):
        lbm.run(1)
        u = get_u(lbm).copy()
        mean = u if mean is None else mean + (u - mean) * np.float32(1.0 / (k + 1))     # Welford's running mean in FP32, like accumulate_from_buffers
    r64 = compare(g, s, get_u(lbm), get_rho(lbm), 64, ceil64, key=fixture + ":u64")
    ravg = compare(g, s, mean, None, 64, ceil64, key=fixture + ":u_avg", field="u_avg")
    return r8, r64, ravg


@pytest.mark.parametrize("case", ["CaseA", "CaseB", "CaseL"])
def test_oracle_fp32_vs_real_reference(case):
    g, s, o = run_case(case, "ref_fp32_%s.npz" % case, False, make_oracle)
    r8, r64, ravg = run_and_compare(o, g, s, "oracle:ref_fp32_%s" % case, 2e-7, 1e-6)
    assert r64 < 1e-5 and ravg < 1e-5                       # the north star's gate, with two orders of margin
    print(case, "fp32 rmse@8 %.3e @64 %.3e u_avg %.3e" % (r8, r64, ravg))


@pytest.mark.parametrize("case", ["CaseA", "CaseL"])
def test_oracle_fp16c_vs_real_reference_shipped_config(case):
    # FP16C storage: 2^-12 relative per stored value.  K = 8 sits inside the north star's 1e-5; at K = 64 the LES case A does NOT (2.6e-5, u_avg 1.5e-5), the
    # laminar case L does (4e-6): stated in bench.py's parity block, gated here at twice the observed values
    g, s, o = run_case(case, "ref_shipped_%s.npz" % case, True, make_oracle)
    r8, r64, ravg = run_and_compare(o, g, s, "oracle:ref_shipped_%s" % case, 2e-6, 6e-5)
    assert r8 < 1e-5
    print(case, "fp16c rmse@8 %.3e @64 %.3e u_avg %.3e" % (r8, r64, ravg))


def test_the_reference_moves_from_itself_by_more_than_the_fp16c_distances_measured_here():
    """The yardstick for every FP16C number (bench line `parity.reference_self_distance_K64`): the reference's FP32 build against its own shipped FP16C build,
    same deck, from the committed fixtures alone.  On the LES case A the storage format alone moves it by 2.2e-5 at K = 64 (1.0e-5 already at K = 8, u_avg
    1.4e-5) -- the 2.6e-5 between this repo's FP16C runs (exact or native arithmetic) and the shipped build are of that size, and no FP16C run, the
    reference's own included, can meet 1e-5 against an FP32 field at that horizon.  Laminar case L: 3.2e-6."""
    import sys
    sys.path.insert(0, os.path.dirname(GOLD[:-len("/golden")]))
    from benchmarks.common import reference_self_distance
    d = reference_self_distance()
    assert 1.5e-5 < d["CaseA"]["K64"] < 3e-5 and 0.5e-5 < d["CaseA"]["K8"] < 1.5e-5 and 1e-5 < d["CaseA"]["u_avg"] < 2e-5
    assert 2e-6 < d["CaseL"]["K64"] < 5e-6
    recorded = json.load(open(os.path.join(GOLD, "observed_rmse.json")))
    ours = recorded["driver:native:ref_shipped_CaseA:final"]
    assert ours < 1.5 * d["CaseA"]["K64"]                      # this repo's FP16C run against the shipped build: the same order as the reference against itself


def test_the_reference_at_c1_size_against_itself():
    """from the fixtures alone (no GPU work): the reference's shipped FP16C build against its own FP32 build on the three mid-planes of the 128^3 deck at
    K = 100 -- 3.2e-5 in lattice units (u_avg 2.2e-5), three times the north star's 1e-5: the yardstick for the FP16C gates above"""
    a, b = np.load(os.path.join(GOLD, "ref_fp32_C1_planes.npz")), np.load(os.path.join(GOLD, "ref_shipped_C1_planes.npz"))
    assert np.array_equal(a["solid_xy"], b["solid_xy"]) and int(a["solid_count"]) == int(b["solid_count"])
    for name, lo, hi in (("u100", 2e-5, 5e-5), ("u_avg", 1e-5, 4e-5)):
        sq, cells = 0.0, 0
        for pl in ("xy", "xz", "yz"):
            fluid = ~a["solid_" + pl]
            d = ((a["%s_%s" % (name, pl)] - b["%s_%s" % (name, pl)]) / (np.float32(7.838) / np.float32(0.1)))[fluid].astype(np.float64)
            sq += float((d ** 2).sum()); cells += int(fluid.sum())
        assert lo < np.sqrt(sq / cells) < hi, (name, np.sqrt(sq / cells))
