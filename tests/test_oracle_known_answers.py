"""Known-answer tests that pin the CPU oracle.  CPU only."""
import numpy as np
import pytest

from oracle import oracle
from helpers import TYPE_S, TYPE_E


def test_fp16c_all_codes_roundtrip_and_range():
    # FX/kernel.cpp:864 comment: range +-1.99951168, smallest normal 6.10351562E-5, smallest denormal 2.98023224E-8
    codes = np.arange(65536, dtype=np.uint32)
    vals = oracle.half_to_float(codes)
    assert np.all(np.isfinite(vals))                       # the format has no Inf/NaN
    back = oracle.float_to_half(vals)
    neg_zero = codes == 0x8000
    assert np.array_equal(back[~neg_zero], codes[~neg_zero].astype(np.uint16))
    assert vals.max() == np.float32(1.99951171875) and vals.min() == np.float32(-1.99951171875)
    pos = vals[(codes < 0x8000) & (codes > 0)]
    assert pos.min() == np.float32(2.98023223876953e-08)
    assert oracle.half_to_float([0x0800])[0] == np.float32(6.103515625e-05)   # smallest normal: e=1, m=0
    # the encoder adds half an ulp (0x800) and truncates the 12 dropped mantissa bits: ties round away from zero
    one = oracle.float_to_half([1.0])[0]
    assert oracle.float_to_half([1.0 + 2.0 ** -13])[0] == one
    assert oracle.float_to_half([1.0 + 2.0 ** -12])[0] == one + 1
    assert oracle.float_to_half([1.0 + 3 * 2.0 ** -12])[0] == one + 2


def test_literal_roundtrip_matches_to_string_float():
    # FX/lbm.cpp:664: def_w is printed with to_string(float) (9 significant digits) and parsed by the compiler
    for x in (1.9999990, 0.00011501, 0.01333333, 1.8368847, 3.0e-8, 123456.78):
        y = oracle.literal(x)
        assert abs(y - np.float32(x)) <= 2 * np.spacing(np.float32(x))
    assert oracle.literal(2.0) == 2.0 and oracle.literal(0.5) == 0.5 and oracle.literal(0.0) == 0.0
    # SURVEY 8c "numerical regime": CaseE nu -> tau 4 ulp above 1/2, def_w printed as 1.99999905
    assert oracle.w_from_nu(7.05e-8) == pytest.approx(1.9999990, abs=2e-7)


def test_rest_state_is_a_fixed_point():
    o = oracle.OracleLBM(12, 10, 8, nu=0.01)
    o.run(5)
    assert not o.fi.any()                                   # DDF-shifting: rest state is all zeros
    assert np.all(o.rho == 1.0) and not o.u.any()


@pytest.mark.parametrize("fp16c", [False, True])
def test_uniform_flow_with_equilibrium_shell_stays_uniform(fp16c):
    Nx, Ny, Nz = 14, 12, 10
    o = oracle.OracleLBM(Nx, Ny, Nz, nu=0.02, fp16c=fp16c)
    f3 = o.flags.reshape(Nz, Ny, Nx)
    f3[0], f3[-1], f3[:, 0], f3[:, -1], f3[:, :, 0], f3[:, :, -1] = (TYPE_E,) * 6
    o.u[:o.N] = 0.05; o.u[o.N:2 * o.N] = -0.02; o.u[2 * o.N:] = 0.01
    o.run(20)
    tol = 2e-4 if fp16c else 2e-7
    assert np.abs(o.u[:o.N] - 0.05).max() < tol and np.abs(o.u[o.N:2 * o.N] + 0.02).max() < tol
    assert np.abs(o.rho - 1.0).max() < tol


def test_mass_and_momentum_conserved_in_periodic_box():
    rng = np.random.default_rng(0)
    o = oracle.OracleLBM(16, 12, 10, nu=0.05)
    o.u[:] = (0.03 * rng.standard_normal(3 * o.N)).astype(np.float32)
    o.rho[:] = (1 + 0.01 * rng.standard_normal(o.N)).astype(np.float32)
    o.initialize()
    rho0, u0 = o.moments()
    m0 = rho0.astype(np.float64).sum(); p0 = (rho0 * u0[:o.N]).astype(np.float64).sum()
    o.run(30)
    rho1, u1 = o.moments()
    assert abs(rho1.astype(np.float64).sum() - m0) / m0 < 1e-6
    assert abs((rho1 * u1[:o.N]).astype(np.float64).sum() - p0) < 1e-3 * max(1.0, abs(p0))


def test_bounce_back_returns_after_exactly_two_steps():
    # solids never touch memory, so a DDF stored INTO a solid neighbour's slot is re-read by the same
    # fluid cell with the opposite direction two steps later (absent after one step).  Nearly collisionless
    # run (w ~ 3e-7) so that the pulse keeps its identity; rho/u are the fields stream_collide writes.
    Nx = 8
    o = oracle.OracleLBM(Nx, 1, 1, nu=1.0e6, subgrid=False)
    o.flags[5] = TYPE_S
    o.initialize()
    pulse = np.float32(0.01)
    o.fi[2 * o.N + 4] = pulse          # plane A(1, t=0) = 2 at cell 4: loaded by cell 4 as f1 (moving +x) at t=0
    o.run(1)
    assert o.rho[4] == pytest.approx(1.0 + pulse, abs=1e-6) and o.u[4] > 0.009       # pulse present, moving +x
    assert o.fi[1 * o.N + 5] == pytest.approx(pulse, rel=1e-5)                         # now sits in the wall cell's slot
    o.run(1)
    assert o.rho[4] == pytest.approx(1.0, abs=1e-6) and abs(o.u[4]) < 1e-6            # absent at t=1
    o.run(1)
    assert o.rho[4] == pytest.approx(1.0 + pulse, abs=1e-6) and o.u[4] < -0.009      # back at t=2, reflected
    assert o.rho[5] == 1.0 and o.u[5] == 0.0                                          # the solid cell is never written


def test_poiseuille_profile_with_volume_force():
    # channel between two solid planes (z), periodic in x,y, driven by fx: u(z) = fx/(2 nu) * (z - z0)(z1 - z) with
    # mid-grid walls half a cell outside the first/last fluid cell (known-answer test 6)
    Nx, Ny, Nz = 4, 4, 18
    nu, fx = 0.1, 1e-5
    o = oracle.OracleLBM(Nx, Ny, Nz, nu=nu, fx=fx, subgrid=False)
    f3 = o.flags.reshape(Nz, Ny, Nx)
    f3[0] = TYPE_S; f3[-1] = TYPE_S
    o.run(6000)
    ux = o.u[:o.N].reshape(Nz, Ny, Nx)[:, 1, 1].astype(np.float64)
    zc = np.arange(Nz) - 0.5              # walls at z = 0.5 and z = Nz - 1.5 in cell-index units
    H = Nz - 2
    ana = fx / (2 * nu) * zc * (H - zc)
    assert np.abs(ux[1:-1] - ana[1:-1]).max() / ana.max() < 0.02


def test_smagorinsky_reduces_to_molecular_rate_at_equilibrium():
    # known-answer 7: with f = feq the strain tensor vanishes, w == def_w and the cell stays at equilibrium
    o = oracle.OracleLBM(6, 6, 6, nu=0.03, subgrid=True)
    o.u[:o.N] = 0.04
    o.initialize()
    before = o.fi.copy()
    o.run(2)
    o2 = oracle.OracleLBM(6, 6, 6, nu=0.03, subgrid=False)
    o2.u[:o2.N] = 0.04
    o2.initialize(); o2.run(2)
    assert np.allclose(o.fi, o2.fi, atol=1e-9) and before.shape == o.fi.shape


def test_halo_extract_insert_roundtrip_two_domains_equals_single_domain():
    # A11: two x-domains with 1-cell halos + extract/exchange/insert == the periodic single domain, bit for bit
    Nx, Ny, Nz = 12, 6, 5
    rng = np.random.default_rng(3)
    u = (0.03 * rng.standard_normal((3, Nz, Ny, Nx))).astype(np.float32)
    rho = (1 + 0.01 * rng.standard_normal((Nz, Ny, Nx))).astype(np.float32)
    flags = np.zeros((Nz, Ny, Nx), np.uint8); flags[1:3, 2:4, 4:9] = TYPE_S
    ref = oracle.OracleLBM(Nx, Ny, Nz, nu=0.01)
    ref.u[:] = u.ravel(); ref.rho[:] = rho.ravel(); ref.flags[:] = flags.ravel()
    ref.run(6)
    h = Nx // 2
    doms = []
    for d in range(2):
        o = oracle.OracleLBM(h + 2, Ny, Nz, nu=0.01, D=(2, 1, 1), O=(d * h - 1, 0, 0))
        xs = (np.arange(-1, h + 1) + d * h) % Nx
        o.u[:] = u[:, :, :, xs].ravel(); o.rho[:] = rho[:, :, xs].ravel(); o.flags[:] = flags[:, :, xs].ravel()
        doms.append(o)
    for o in doms: o.initialize()
    # LBM::initialize communicates fi once with odd t (FX/lbm.cpp:1242-1252)
    def exchange(t):
        ex = [o.extract_fi(0, t) for o in doms]
        for d, o in enumerate(doms):
            # domain d receives into its +halo what its +x neighbour extracted on its - side, and vice versa
            o.insert_fi(0, ex[(d + 1) % 2][1], ex[(d - 1) % 2][0], t)
    exchange(1)
    for step in range(6):
        for o in doms: o.stream_collide()
        exchange(doms[0].t)
        for o in doms: o.t += 1
    got = np.zeros((3, Nz, Ny, Nx), np.float32)
    for d, o in enumerate(doms):
        got[:, :, :, d * h:(d + 1) * h] = o.u.reshape(3, Nz, Ny, h + 2)[:, :, :, 1:-1]
    assert np.array_equal(got, ref.u.reshape(3, Nz, Ny, Nx))
