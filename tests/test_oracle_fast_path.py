"""The oracle's row-wise path (eight cells of a row per SIMD statement, indices by addition, ramps from tables) against its literal path (one call of
stream_collide_cell per cell, the reference's own structure): the SAME BITS in rho, u, T and every stored DDF -- FP32 and FP16C, every force term, thermal
lattice, halo'ed domains of every split, odd sizes whose last chunk is ragged, both time parities -- and on the states of the committed reference cases."""
import os

import numpy as np
import pytest

from helpers import synthetic_state, thermal_state
from oracle import oracle, setup_profile

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def run_both(make, steps=5):
    out = []
    for fast in (False, True):
        oracle.set_fast(fast)
        try:
            o = make()
            o.run(steps)
            out.append(o)
        finally:
            oracle.set_fast(True)
    a, b = out
    for name in ("fi", "rho", "u") + (("gi", "T") if a.thermal else ()):
        assert np.array_equal(bits(getattr(a, name)), bits(getattr(b, name))), name + ": the row-wise path differs from the literal path"
    assert np.abs(a.u).max() > 0


pytestmark = pytest.mark.skipif(not oracle.fast_available(), reason="the oracle was built without its row-wise path (no AVX2 + FMA)")


@pytest.mark.parametrize("fp16c", [False, True])
@pytest.mark.parametrize("size", [(48, 40, 24), (37, 9, 5), (8, 6, 4), (23, 5, 7), (9, 4, 3)])
def test_rows_equal_cells_with_every_force_term(size, fp16c):
    Nx, Ny, Nz = size
    st = synthetic_state(Nx, Ny, Nz, seed=3, shell="luw")
    rng = np.random.default_rng(7)
    F = (1e-5 * rng.standard_normal(3 * Nx * Ny * Nz)).astype(np.float32)

    def make():
        o = oracle.OracleLBM(Nx, Ny, Nz, 2e-5, 1e-5, -2e-5, 3e-6, fp16c=fp16c, use_F=True)
        o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]; o.F[:] = F
        o.set_coriolis(0.0, 3e-5, 4e-5)
        o.set_buffer_nudging(max(1, min(5, Nx // 4)), 0.0133333, 2, 1); o.set_top_sponge(max(1, min(6, Nz // 2)), 0.02)
        return o
    run_both(make, steps=7)


@pytest.mark.parametrize("shell", [None, "E"])
def test_rows_equal_cells_periodic_and_plain(shell):
    Nx, Ny, Nz = 22, 7, 6
    st = synthetic_state(Nx, Ny, Nz, seed=9, solids=shell is not None, shell=shell)

    def make():
        o = oracle.OracleLBM(Nx, Ny, Nz, 1e-4, subgrid=shell is not None)
        o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
        return o
    run_both(make, steps=6)


@pytest.mark.parametrize("fp16c", [False, True])
@pytest.mark.parametrize("sponge", [False, True])
def test_rows_equal_cells_thermal(fp16c, sponge):
    Nx, Ny, Nz = 27, 10, 9
    st = synthetic_state(Nx, Ny, Nz, seed=4, shell="luw")
    fl, T = thermal_state(st[0], (Nx, Ny, Nz))
    # temperature boundary on the velocity boundary: the top layer the sponge reads is preset (no cell rewrites it mid-step)
    fl = fl.copy(); fl[(fl & 3) == 2] |= 4

    def make():
        o = oracle.OracleLBM(Nx, Ny, Nz, 2e-5, fp16c=fp16c, alpha=2e-5)
        o.flags[:] = fl; o.u[:] = st[1]; o.rho[:] = st[2]; o.T[:] = T
        if sponge:
            o.set_top_sponge(3, 0.02)
        return o
    run_both(make, steps=6)


@pytest.mark.parametrize("D,O", [((2, 1, 1), (-1, 0, 0)), ((2, 2, 1), (15, -1, 0)), ((1, 2, 2), (0, 7, 5)), ((2, 2, 2), (15, 7, -1))])
def test_rows_equal_cells_on_haloed_domains(D, O):
    # one domain of a split lattice, in its local shape (halo layers on the split axes): local zones, halo cells skipped
    Nx, Ny, Nz = 18, 10, 8
    st = synthetic_state(Nx, Ny, Nz, seed=6, shell=None)
    # the global outer faces this domain owns become TYPE_E: the cells the nudging / sponge terms read are inputs (a fluid cell there is rewritten by its own
    # thread while others read it -- in the reference's kernel as in either path here -- and the outcome depends on who comes first)
    fl = st[0].reshape(Nz, Ny, Nx).copy()
    gN = [(n - 2 * (d > 1)) * d for n, d in zip((Nx, Ny, Nz), D)]
    for axis, (n, o_, g) in enumerate(zip((Nx, Ny, Nz), O, gN)):
        for face in (-o_, g - 1 - o_):
            if 0 <= face < n:
                sl = [slice(None)] * 3; sl[2 - axis] = face
                fl[tuple(sl)] = np.where(fl[tuple(sl)] == 1, 1, 2)
    fl = fl.ravel()

    def make():
        o = oracle.OracleLBM(Nx, Ny, Nz, 2e-5, fp16c=True, D=D, O=O)
        o.flags[:] = fl; o.u[:] = st[1]; o.rho[:] = st[2]
        o.set_buffer_nudging(3, 0.0133333, 2, 0); o.set_top_sponge(3, 0.02); o.set_coriolis(0.0, 3e-5, 4e-5)
        return o
    run_both(make, steps=4)


@pytest.mark.parametrize("case,fp16c", [("CaseA", False), ("CaseA", True), ("CaseB", False), ("CaseL", True)])
def test_rows_equal_cells_on_the_reference_cases(case, fp16c):
    g = np.load(os.path.join(GOLD, ("ref_shipped_%s.npz" if fp16c and case != "CaseB" else "ref_fp32_%s.npz") % case))
    s = setup_profile.setup_profile_case(os.path.join(GOLD, "refcases", case, "conf.luwpf"), solid_mask=g["solid"])
    from test_oracle_vs_reference import make_oracle
    run_both(lambda: make_oracle(s, fp16c), steps=16)
