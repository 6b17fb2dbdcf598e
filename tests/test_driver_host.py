"""Host stage of the C++ driver (latticeurbanwind_amd/host/luw_driver, the `FluidX3D <deck>` replacement) on CPU:
derived quantities against numbers printed by the REAL reference binary (tests/golden/*.console.txt and SURVEY.md 8c
host-stage goldens), and the initial lattice state against the Python restatement (oracle/setup_profile.py)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import setup_profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(ROOT, "latticeurbanwind_amd", "host", "luw_driver")


@pytest.fixture(scope="module")
def driver(luw):
    subprocess.check_call(["make", "-C", os.path.dirname(DRIVER), "-s"])
    return DRIVER


def run(driver, deck, *args):
    r = subprocess.run([driver, deck, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def read_dump(path):
    raw = open(path, "rb").read()
    Nx, Ny, Nz, Nzc = struct.unpack_from("<4I", raw, 0)
    fh = struct.unpack_from("<8f", raw, 16); ih = struct.unpack_from("<8i", raw, 48)
    N = Nx * Ny * Nz
    flags = np.frombuffer(raw, np.uint8, N, 80)
    u = np.frombuffer(raw, np.float32, 3 * N, 80 + N)
    extra = {}
    off = 80 + N + 12 * N
    if raw[off:off + 4] == b"TEMP":      # temperature cases: unit_K, unit_K_offset, T[N] in lattice units
        extra = dict(unit_K=np.frombuffer(raw, np.float32, 1, off + 4)[0], unit_K_offset=np.frombuffer(raw, np.float32, 1, off + 8)[0],
            T=np.frombuffer(raw, np.float32, N, off + 12))
    return dict(extra, Nx=Nx, Ny=Ny, Nz=Nz, Nz_core=Nzc, nu=fh[0], si_u=fh[1], si_rho=fh[2], buffer_inv_tau=fh[3], sponge_inv_tau=fh[4], scale=fh[5],
                buffer_active=ih[0], buffer_N=ih[1], buffer_face=ih[2], nudge_vertical=ih[3], sponge_active=ih[4], sponge_N=ih[5], nvox=ih[6], mapped=ih[7],
                    flags=flags, u=u)


@pytest.mark.parametrize("case", ["CaseA", "CaseB", "CaseL"])
def test_initial_state_equals_python_restatement(driver, tmp_path, case):
    deck = os.path.join(GOLD, "refcases", case, "conf.luwpf")
    dump = str(tmp_path / "setup.bin")
    out = run(driver, deck, "--dry-run", "--dump-setup", dump)
    d = read_dump(dump)
    s = setup_profile.setup_profile_case(deck)
    assert (d["Nx"], d["Ny"], d["Nz"], d["Nz_core"]) == (s["Nx"], s["Ny"], s["Nz"], s["Nz_core"])
    assert np.float32(d["nu"]) == np.float32(s["nu"]) and np.float32(d["si_u"]) == np.float32(s["si_u_factor"]) and np.float32(d["si_rho"]) == np.float32(
        s["si_rho_factor"])
    assert bool(d["buffer_active"]) == bool(s["buffer_active"]) and d["buffer_N"] == s["buffer_N"] and d["buffer_face"] == s["buffer_face"]
    assert np.float32(d["buffer_inv_tau"]) == np.float32(s["buffer_inv_tau"]) and np.float32(d["sponge_inv_tau"]) == np.float32(s["sponge_inv_tau"])
    assert bool(d["sponge_active"]) == bool(s["sponge_active"]) and d["sponge_N"] == s["sponge_N"]
    assert d["nvox"] == s["n_solid_vox"] and d["mapped"] == s["mapped_bc"]
    assert np.array_equal(d["flags"], s["flags"]) and np.array_equal(d["u"], s["u"])
    # and the console lines the real reference printed for the same deck
    ref = open(os.path.join(GOLD, "ref_fp32_%s.console.txt" % case)).read()
    for key in ("Grid Resolution", "Top sponge grid", "Profile table", "Profile U range", "Profile z range", "STL bounds SI"):
        mine = [l for l in out.splitlines() if key in l]
        theirs = [l for l in ref.splitlines() if key in l]
        assert bool(mine) == bool(theirs) and all(" ".join(l.split()) in [" ".join(t.split()) for t in theirs] for l in mine), key
    for frag in ("Unit Conversion: 1 cell =", "Nbuf=", "inv_tau_lbmu=", "Nsponge=", "scaled by", "profile boundaries mapped:", "below local terrain"):
        mine = sorted(" ".join(l.split("|")[-2 if l.rstrip().endswith("|") else -1].split()) for l in out.splitlines() if frag in l)
        theirs = sorted(" ".join(l.split("|")[-2].split()) for l in ref.splitlines() if frag in l)
        assert bool(mine) == bool(theirs) and all(m in theirs for m in mine), (frag, mine, theirs)


def write_case_e_like(dirpath, deck_lines):
    """a project with the numbers of the reference's CaseE example: our own files, same magnitudes"""
    import sys
    sys.path.insert(0, GOLD)
    from make_refcases import box_tris, write_stl
    os.makedirs(os.path.join(dirpath, "proj_temp")); os.makedirs(os.path.join(dirpath, "wind_bc"))
    write_stl(os.path.join(dirpath, "proj_temp", "CaseE_PF.stl"), box_tris(0, 2022.500153, 0, 1996.500092, -20.0, 0.0)
        + box_tris(900.0, 950.0, 900.0, 960.0, 0.0, 59.9))
    with open(os.path.join(dirpath, "wind_bc", "profile.dat"), "w") as f:
        f.write("z,U\n" + "\n".join("%g\t%g" % zu for zu in [(1.25, 2.847), (5, 3.26), (25, 4.36), (100, 6.16), (250, 7.8)]) + "\n")
    base = ["casename = CaseE", "datetime = 20251222120000", "si_x_cfd = [0.000000, 2022.500153]", "si_y_cfd = [0.000000, 1996.500092]",
            "si_z_cfd = [0.000000, 270.000000]", "base_height = 20.0", "validation = pass", "angle = [0, 90, 180, 270]"]
    deck = os.path.join(dirpath, "conf.luwpf")
    open(deck, "w").write("\n".join(base + deck_lines) + "\n")
    return deck


# host-stage goldens captured from the real reference binary: deck edits -> grid; unit line; Nbuf/inv_tau; Nsponge/inv_tau
SIZING = [
    (['n_gpu = [2, 1, 1]', 'mesh_control = "gpu_memory"', 'gpu_memory = 4000'], "751,  742,  174 (nCell = 96960108)", "2x 3997 MB (core 3960 + extra 37)",
     "1 cell = 2690.701 mm, 1 s = 29 time steps", "Nbuf=43 cells", "inv_tau_lbmu=0.00011501", "Nsponge=74 cells", "inv_tau_lbmu=0.00028753",
         "side_ref_cap_z=99"),
    (['n_gpu = [1, 1, 1]', 'mesh_control = "cell_size"', 'cell_size = 16'], "126,  125,   30", None,
     "1 cell = 15972.001 mm, 1 s = 5 time steps", "Nbuf=7 cells", "inv_tau_lbmu=0.00068376", "Nsponge=13 cells", "inv_tau_lbmu=0.00170940", None),
    (['n_gpu = [1, 1, 1]', 'mesh_control = "cell_size"', 'cell_size = 8', 'enable_top_sponge = false'], "253,  250,   34", None,
     "1 cell = 7986.000 mm, 1 s = 10 time steps", "Nbuf=8 cells", "inv_tau_lbmu=0.00034188", None, "inv_tau_lbmu=0.00085470", None),
    (['n_gpu = [1, 1, 1]', 'mesh_control = "gpu_memory"', 'gpu_memory = 20000'], "1023, 1009,  238", "1x 19994 MB (core 19925 + extra 69)",
     "1 cell = 1978.692 mm, 1 s = 39 time steps", "Nbuf=59 cells", "inv_tau_lbmu=0.00008452", "Nsponge=101 cells", "inv_tau_lbmu=0.00021129", None),
    (['n_gpu = [4, 2, 1]', 'mesh_control = "gpu_memory"', 'gpu_memory = 40000', 'turb_inflow_enable = false'], "2579, 2545,  599 (nCell = 3931569445)",
        "8x 39996 MB",
     "1 cell = 784.479 mm, 1 s = 99 time steps", "Nbuf=149 cells", "inv_tau_lbmu=0.00003352", "Nsponge=255 cells", "inv_tau_lbmu=0.00008380", None),
    (['n_gpu = [2, 2, 1]', 'mesh_control = "cell_size"', 'cell_size = 5', 'enable_top_sponge = false', 'enable_buffer_nudging = false',
        'turb_inflow_enable = false'],
     "405,  399,   54", "4x 191 MB", "1 cell = 5003.760 mm, 1 s = 16 time steps", "Nbuf=13 cells", "inv_tau_lbmu=0.00021368", None, None, None),
]


@pytest.mark.parametrize("idx", range(len(SIZING)))
def test_grid_sizing_against_reference_host_stage_goldens(driver, tmp_path, idx):
    lines, *expect = SIZING[idx]
    deck = write_case_e_like(str(tmp_path / "proj"), lines)
    out = " ".join(run(driver, deck, "--sizing-only").split())
    for e in expect:
        if e is not None:
            assert " ".join(e.split()) in out, (e, out[-1500:])


def test_deck_grammar(driver, tmp_path):
    # FX/setup.cpp:61-178: comments outside quotes, key normalisation, fuzzy bools, last-wins
    deck = write_case_e_like(str(tmp_path / "p"), ['N-GPU = [1, 1, 1]   // trailing comment', 'Mesh Control = "cell_size"', 'cell_size = 99', 'CELL_SIZE = 16',
                                                    'enable top sponge = "OFF"', 'enable_buffer_nudging = 0.0', 'vk_inlet_enable = no'])
    out = " ".join(run(driver, deck, "--sizing-only").split())
    assert "126, 125, 17" in out and "Top sponge | disabled" in out and "Buffer nudging | disabled" in out and "von-Karman" not in out


def test_luw_deck_without_surfdata_fails_loudly(driver, tmp_path):
    deck = str(tmp_path / "x.luw"); open(deck, "w").write(
        "casename = x\ndatetime = 20260101120000\nsi_x_cfd = [0, 96]\nsi_y_cfd = [0, 80]\nsi_z_cfd = [0, 48]\n")
    r = subprocess.run([driver, deck, "--dry-run"], capture_output=True, text=True)
    assert r.returncode != 0 and "could not open CSV" in r.stdout and "no inlet samples" in r.stdout


@pytest.mark.parametrize("case", ["CaseN1", "CaseN2", "CaseN3", "CaseN4"])
def test_nwp_boundary_builders_against_real_reference_fields(driver, tmp_path, case):
    """*.luw host stage on CPU (SurfData reader, patch-driven 2-D mapping / KNN-HD / nearest-sample fill, terrain clip, flux
    correction): TYPE_E cells keep the velocity the builders wrote, so the first u output of the REAL reference shows their
    result; the driver's initial state must equal it bit for bit on the side faces.  (--dry-run voxelises with the IEEE host
    restatement, whose mask can differ from the device's on lattice-plane faces: cells solid in either mask are skipped.)"""
    deck = os.path.join(GOLD, "refcases", case, "conf.luw")
    dump = str(tmp_path / "setup.bin")
    out = run(driver, deck, "--dry-run", "--dump-setup", dump)
    d = read_dump(dump)
    gold = np.load(os.path.join(GOLD, "ref_fp32_%s.npz" % case))
    Nx, Ny, Nz, Nzc = d["Nx"], d["Ny"], d["Nz"], d["Nz_core"]
    assert tuple(gold["dims"]) == (Nx, Ny, Nzc)
    u = (d["u"].reshape(3, Nz, Ny, Nx)[:, :Nzc] * np.float32(d["si_u"])).astype(np.float32).transpose(1, 2, 3, 0)
    fl = d["flags"].reshape(Nz, Ny, Nx)[:Nzc]
    side = np.zeros(fl.shape, bool); side[:, 0, :] = side[:, -1, :] = side[:, :, 0] = side[:, :, -1] = True
    if Nzc == Nz: side[-1] = True                                     # the top face too when it is part of the output
    m = side & ((fl & 1) == 0) & ~gold["solid"]
    assert m.sum() > 3000 and np.array_equal((fl[m] & 2), np.full(int(m.sum()), 2, np.uint8))
    assert np.array_equal(u[m], gold["u8"][m]), "%d boundary cells differ" % int((u[m] != gold["u8"][m]).any(-1).sum())
    # console numbers of the reference's host stage
    ref = open(os.path.join(GOLD, "ref_fp32_%s.console.txt" % case)).read()
    norm = lambda txt: [" ".join(l.strip().strip("|").split()) for l in txt.splitlines()]
    mine_all, theirs_all = norm(out), norm(ref)
    for frag in ("Unit Conversion: 1 cell =", "CDF data loaded", "S_in=", "corrected=", "per-face dU", "patch-driven 2D mapping", "bottom =", "top =",
            "south =", "north =", "west =", "east ="):
        mine = sorted(l for l in mine_all if frag in l)
        theirs = sorted(l for l in theirs_all if frag in l)
        assert mine == theirs, (frag, mine, theirs)


def test_probe_requests_resolve_like_the_reference(driver, tmp_path):
    """deck key `probes`: token grammar (centre / quoted centre + grid offset / metre offset / lon:lat / lon:lat + offset /
    outside the domain), WGS84 -> UTM -> rotated local frame, snapping to a lattice column, file stems: the rows the REAL
    reference printed for the same deck (levels depend on the voxel mask: host-voxelised here, so only stems and columns
    are compared on the CPU; tests/test_gpu_driver.py compares the CSV files)"""
    out = run(driver, os.path.join(GOLD, "refcases", "CaseP", "conf.luwpf"), "--dry-run")
    ref = open(os.path.join(GOLD, "ref_fp32_CaseP.console.txt")).read()
    import re
    pick = lambda txt: re.findall(r"(\S+) -> \((\d+),(\d+)\), levels=", txt)
    assert pick(out) == pick(ref) and len(pick(ref)) == 5
    norm = lambda txt: [" ".join(l.strip().strip("|").split()) for l in txt.splitlines() if "ignored:" in l or "Probes Window" in l or "request(s)" in l]
    assert norm(out) == norm(ref)


@pytest.mark.parametrize("case", ["CaseT1", "CaseT2", "CaseT3"])
def test_temperature_boundaries_against_real_reference_fields(driver, tmp_path, case):
    """buoyancy = true + a T column in the CSV: adaptive Kelvin <-> lattice map, patch-driven / KNN-HD / nearest-sample temperature
    on the boundary cells, ground-temperature plane on the solids.  TYPE_T cells keep their preset, so the reference's final T
    output shows the builders' result there: the driver's initial T must equal it bit for bit (cells whose solid state differs
    between the host voxeliser of --dry-run and the device mask are skipped)."""
    deck = os.path.join(GOLD, "refcases", case, "conf.luw")
    dump = str(tmp_path / "setup.bin")
    out = run(driver, deck, "--dry-run", "--dump-setup", dump)
    d = read_dump(dump)
    gold = np.load(os.path.join(GOLD, "ref_fp32_%s.npz" % case))
    Nx, Ny, Nz, Nzc = d["Nx"], d["Ny"], d["Nz"], d["Nz_core"]
    fl = d["flags"].reshape(Nz, Ny, Nx)[:Nzc]
    T_si = (d["T"].reshape(Nz, Ny, Nx)[:Nzc] * d["unit_K"] + d["unit_K_offset"]).astype(np.float32)
    preset = (fl & 4) != 0
    same_solid = ((fl & 1) != 0) == gold["solid"]
    m = preset & same_solid
    assert m.sum() > 2500 and np.array_equal(T_si[m], gold["T16"][m]), "%d preset cells differ" % int((T_si[m] != gold["T16"][m]).sum())
    ref = open(os.path.join(GOLD, "ref_fp32_%s.console.txt" % case)).read()
    norm = lambda txt, keys: sorted(" ".join(l.strip().strip("|").split()) for l in txt.splitlines() if any(k in l for k in keys))
    keys = ("Temp Reference", "Temp Scale", "Thermal alpha", "Thermal tau_T", "Thermal beta", "T patch", "CSV T range", "T column")
    assert norm(out, keys) == norm(ref, keys)
