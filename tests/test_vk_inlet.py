"""Von-Karman synthetic-turbulence inlet.  CPU: the restatement (own mt19937_64 / generate_canonical, mode
spectrum, cell selection, kernel) against fields of the REAL reference run with turb_inflow_enable=true, and the product's
C++ table builder (host/vk_inlet.hpp through the driver) against the restatement.  GPU: the device kernel + run-loop
integration against the restatement and the real reference."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle, setup_profile, vk_inlet
from test_oracle_vs_reference import compare, make_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")
DECK = os.path.join(GOLD, "refcases", "CaseV", "conf.luwpf")


def case_v(npz, solid=True):
    g = np.load(os.path.join(GOLD, npz))
    s = setup_profile.setup_profile_case(DECK, solid_mask=g["solid"] if solid else None)
    L_lbm = s["units"].x(np.float32(20.0))                      # vk_inlet_l = 20 m
    T = vk_inlet.build_tables(s["Nx"], s["Ny"], s["Nz"], s["flags"], s["u"], ti=0.05, L_lbm=L_lbm, nmodes=64, seed=100,
                              face_mode="ALL_SIDES", downstream_face_id=1)
    return g, s, T


def test_mt19937_64_known_answer():
    # ISO C++ [rand.predef]: the 10000th consecutive invocation of a default-constructed mt19937_64 is 9981545732273789042
    r = vk_inlet.MT19937_64(5489)
    for _ in range(9999): r()
    assert r() == 9981545732273789042


def test_mode_basis_has_unit_rms():
    m = vk_inlet.build_modes_for_seed(25.0, 256, 0.08, (1.0, 0.0, 0.0), 100)
    assert abs(0.5 * float((m[:, 4].astype(np.float64) ** 2).sum()) - 1.0) < 1e-5      # sum A^2/2 = 1 (FX/setup.cpp:834-848)
    k = np.sqrt((m[:, :3].astype(np.float64) ** 2).sum(1))
    assert k.min() >= 2 * np.pi / 250 * 0.999 and k.max() <= np.pi * 1.001              # k in [2 pi/(10 L), pi]
    assert np.allclose(m[:, 3], 0.08 * m[:, 0], rtol=1e-6)                             # omega = u_ref k.conv_dir


@pytest.mark.parametrize("npz,fp16c,g8,g64", [("ref_fp32_CaseV.npz", False, 2e-7, 1e-6), ("ref_shipped_CaseV.npz", True, 2e-6, 1e-4)])
def test_oracle_with_vk_inlet_vs_real_reference(npz, fp16c, g8, g64):
    g, s, T = case_v(npz)
    txt = open(os.path.join(GOLD, npz.replace(".npz", ".console.txt"))).read()
    assert "west: points=800, Uc=0.082869" in txt and "south: points=920, Uc=0.082870" in txt and T["point_count"] == 3440
    o = make_oracle(s, fp16c)
    o.initialize()
    for t in range(64):
        oracle.vk_inlet_apply(o, T, *vk_inlet.time_params(o.t))
        o.run(1)
        if o.t == 8: compare(g, s, o.u, None, 8, g8, key="oracle:%s:u8" % npz[:-4])
    compare(g, s, o.u, o.rho, 64, g64, key="oracle:%s:u64" % npz[:-4])


def test_driver_vk_tables_equal_restatement(luw, tmp_path):
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    dump = str(tmp_path / "vk.bin")
    r = subprocess.run([DRIVER, DECK, "--dry-run", "--dump-vk", dump], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "west: points=800" in r.stdout and "north: points=920" in r.stdout
    raw = open(dump, "rb").read()
    P, M = struct.unpack_from("<2Q", raw, 0)
    off = 16
    cell = np.frombuffer(raw, np.uint64, P, off); off += 8 * P
    face = np.frombuffer(raw, np.uint8, P, off); off += P
    pdata = np.frombuffer(raw, np.float32, 7 * P, off); off += 28 * P
    mdata = np.frombuffer(raw, np.float32, 50 * M, off)
    s = setup_profile.setup_profile_case(DECK)                 # the driver voxelises itself, so does the restatement
    T = vk_inlet.build_tables(s["Nx"], s["Ny"], s["Nz"], s["flags"], s["u"], ti=0.05, L_lbm=s["units"].x(np.float32(20.0)), nmodes=64, seed=100,
                              face_mode="ALL_SIDES", downstream_face_id=1)
    assert (P, M) == (T["point_count"], T["mode_count"])
    assert np.array_equal(cell, T["point_cell"]) and np.array_equal(face, T["point_face"])
    assert np.array_equal(pdata, T["point_data"]) and np.array_equal(mdata, T["mode_data"])


def test_time_params_stride_and_interpolation():
    assert vk_inlet.time_params(7) == (0, 7.0, 7.0, 0.0)
    assert vk_inlet.time_params(7, 4, False) == (0, 4.0, 4.0, 0.0)
    ui, t0, t1, a = vk_inlet.time_params(7, 4, True)
    assert (ui, t0, t1) == (1, 4.0, 8.0) and a == np.float32(0.75)


@pytest.mark.gpu
@pytest.mark.parametrize("stride,interp", [(1, False), (3, True)])
def test_hip_vk_inlet_matches_restatement(luw, stride, interp):
    # device cosf (ocml) vs glibc cosf differ in the last bit now and then: tolerance instead of equality (stated: 5e-7 on
    # the inlet velocities, 2e-6 RMSE on the whole field after 24 LES steps)
    g, s, T = case_v("ref_fp32_CaseV.npz")
    lbm = luw.LBM(s["Nx"], s["Ny"], s["Nz"], float(s["nu"]), update_fields_every_step=True)
    lbm.flags.data[:] = s["flags"]; lbm.u.data[:] = s["u"]; lbm.rho.data[:] = s["rho"]
    o = make_oracle(s, False)
    lbm.run(0); o.initialize()
    lbm.vk_inlet_attach(T["point_cell"], T["point_face"], T["point_data"], T["mode_data"], T["mode_count"], stride, interp)
    lbm.vk_inlet_apply(); lbm.finish(); lbm.u.read_from_device()
    oracle.vk_inlet_apply(o, T, *vk_inlet.time_params(0, stride, interp))
    assert np.abs(lbm.u.data - o.u).max() < 5e-7
    for t in range(24):
        oracle.vk_inlet_apply(o, T, *vk_inlet.time_params(o.t, stride, interp))
        o.run(1)
    lbm.run(24)
    lbm.u.read_from_device()
    fluid = (s["flags"] & 1) == 0
    d = (lbm.u.data.reshape(3, -1) - o.u.reshape(3, -1))[:, fluid].astype(np.float64)
    assert np.sqrt((d ** 2).sum(0).mean()) < 2e-6


@pytest.mark.gpu
def test_hip_vk_inlet_vs_real_reference(luw):
    g, s, T = case_v("ref_fp32_CaseV.npz")
    lbm = luw.LBM(s["Nx"], s["Ny"], s["Nz"], float(s["nu"]))
    lbm.flags.data[:] = s["flags"]; lbm.u.data[:] = s["u"]; lbm.rho.data[:] = s["rho"]
    lbm.run(0)
    lbm.vk_inlet_attach(T["point_cell"], T["point_face"], T["point_data"], T["mode_data"], T["mode_count"])
    lbm.run(8); lbm.u.read_from_device()
    # (the device library's cosf against the host's: the inlet values may differ in the last bit, so this run has its own record)
    compare(g, s, lbm.u.data, None, 8, 2e-7, key="hip:ref_fp32_CaseV:u8")
    lbm.run(56); lbm.u.read_from_device(); lbm.rho.read_from_device()
    compare(g, s, lbm.u.data, lbm.rho.data, 64, 1e-6, key="hip:ref_fp32_CaseV:u64")


@pytest.mark.gpu
def test_values_computed_ahead_equal_values_computed_in_line(luw, tmp_path):
    """the product evaluates the inlet values of step t+1 on a side stream while step t runs and scatters them into u before the
    next step; LUW_TEST_AIDS=vk_inline evaluates them in line in front of every step.  Same kernel arithmetic: the deck driver must write
    byte-identical files either way (case V: inlet on, unsteady outputs, averaging; and with update stride 3 + interpolation)."""
    import filecmp, glob, shutil, subprocess
    driver = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")
    subprocess.check_call(["make", "-C", os.path.dirname(driver), "-s"])
    for tag, extra in (("plain", ""), ("stride", "vk_inlet_update_stride = 3\nvk_inlet_stride_interpolation = true\n")):
        out = {}
        for mode in ("1", "0"):
            proj = str(tmp_path / (tag + mode))
            shutil.copytree(os.path.join(GOLD, "refcases", "CaseV"), proj)
            deck = os.path.join(proj, "conf.luwpf")
            open(deck, "a").write("\n" + extra)
            r = subprocess.run([driver, deck, "--ddf", "fp32"], capture_output=True, text=True, timeout=600,
                env=dict(os.environ, LUW_TEST_AIDS=("" if mode == "1" else "vk_inline")))
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
            out[mode] = sorted(glob.glob(os.path.join(proj, "RESULTS", "vtk", "*.vtk")))
        assert len(out["1"]) >= 4 and [os.path.basename(p) for p in out["1"]] == [os.path.basename(p) for p in out["0"]]
        for a, b in zip(out["1"], out["0"]):
            assert filecmp.cmp(a, b, shallow=False), (tag, os.path.basename(a))
