"""Parity of the HIP path (through the C-ABI) against the CPU oracle on identical inputs.  GPU only.

Bar: value-exact.  The kernels and the oracle implement one arithmetic contract (FP32, fmaf where the reference
writes fma, every other op rounded separately, IEEE / and sqrt), so rho, u and all 19 DDF planes must compare equal
element for element (np.array_equal; +0 == -0) -- for FP32 and for FP16C storage, for both kernels."""
import os

import numpy as np
import pytest

from helpers import synthetic_state, TYPE_S, TYPE_E, TYPE_T

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_pair(luw, oracle, Nx, Ny, Nz, nu, fp16c, kernel, state, force=(0, 0, 0), coriolis=None, nudging=None, sponge=None,
              use_F=False, every_step=False, subgrid=True):
    from latticeurbanwind_amd import capi
    flags, u, rho = state
    g = luw.LBM(Nx, Ny, Nz, nu, *force, fp16c=fp16c, kernel={"auto": capi.KERNEL_AUTO, "s": capi.KERNEL_SCALAR, "p": capi.KERNEL_PAIR}[kernel],
                force_field=use_F, update_fields_every_step=every_step, subgrid=subgrid,
                buffer_nudging=nudging, top_sponge=sponge)
    o = oracle.OracleLBM(Nx, Ny, Nz, nu, *force, fp16c=fp16c, use_F=use_F, subgrid=subgrid)
    for l in (g, o):
        fl = l.flags.data if hasattr(l.flags, "data") else l.flags
        fl[:] = flags
        (l.u.data if hasattr(l.u, "data") else l.u)[:] = u
        (l.rho.data if hasattr(l.rho, "data") else l.rho)[:] = rho
    if use_F:
        rng = np.random.default_rng(7)
        F = (1e-5 * rng.standard_normal(3 * Nx * Ny * Nz)).astype(np.float32)
        g.F.data[:] = F; o.F[:] = F
    if coriolis:
        g.set_coriolis(*coriolis); o.set_coriolis(*coriolis)
    if nudging:
        o.set_buffer_nudging(nudging["n_cells"], nudging["inv_tau"], nudging.get("downstream_face", 0), nudging.get("nudge_vertical", 0))
    if sponge:
        o.set_top_sponge(sponge["n_cells"], sponge["inv_tau"])
    return g, o


def ddf_equal(fi, ref):
    """value equality; FP16C codes 0x0000/0x8000 are both zero"""
    if fi.dtype == np.uint16:
        a = fi.copy(); b = ref.copy()
        a[a == 0x8000] = 0; b[b == 0x8000] = 0
        return np.array_equal(a, b)
    return np.array_equal(fi, ref)


def check(g, o, what):
    g.u.read_from_device(); g.rho.read_from_device()
    fi = g.download_fi()
    nbad = int((fi != o.fi).sum())
    assert ddf_equal(fi, o.fi), "%s: %d DDF values differ" % (what, nbad)
    assert np.array_equal(g.rho.data, o.rho), what + ": rho differs"
    assert np.array_equal(g.u.data, o.u), what + ": u differs"


def test_fp16c_codec_exhaustive(luw):
    # the kernels' fast FP16C codec vs the literal FX/kernel.cpp:864-875 formulas: all 2^16 codes, all 2^32 floats
    import ctypes as C
    from latticeurbanwind_amd import capi
    n = C.c_uint64(123)
    capi.check(capi.load().luw_selfcheck_fp16c_codec(0, C.byref(n)))
    assert n.value == 0


def test_plain_range_division_and_square_root_equal_the_library_forms(luw):
    # recip_prepare / div_by / sqrt_in_range (csrc/luw_device.hpp) vs `a/b` and sqrtf() on the device: every float of the square root's range,
    # 2^31 quotients over every denominator of [1/4, 4], and the same on the 2^-25 grid of FP16C moment sums
    import ctypes as C
    from latticeurbanwind_amd import capi
    n = (C.c_uint64 * 3)(7, 7, 7)
    capi.check(capi.load().luw_selfcheck_arith(0, n))
    assert list(n) == [0, 0, 0]


def test_density_outside_the_plain_range_takes_the_library_path(luw):
    """densities outside [1/4, 4] (nothing a healthy lattice holds): the FP16C kernels vote per wave and fall back to the library's division and
    square root -- the oracle's IEEE results, bit for bit, in both regimes and in waves that mix them"""
    from oracle import oracle
    Nx, Ny, Nz = 512, 6, 5
    flags, u, rho = synthetic_state(Nx, Ny, Nz, seed=77, shell="luw")
    rho = rho.copy().reshape(Nz, Ny, Nx)
    rho[:, 2, :] = 0.2; rho[:, 3, 100:300] = 4.5; rho[2, 4, 7] = 0.01     # whole rows, part of a row, one cell
    for kern in ("p", "s"):
        g, o = make_pair(luw, oracle, Nx, Ny, Nz, 1e-4, True, kern, (flags, u, rho.ravel()), coriolis=(0.0, 3e-5, 4e-5))
        g.run(3); o.run(3)
        check(g, o, "densities outside the plain range, kernel " + kern)
        g.close()


SIZES = [(32, 32, 32), (48, 40, 24), (37, 19, 11), (6, 5, 7), (3, 4, 5), (130, 6, 5), (260, 3, 4), (2, 3, 3), (514, 4, 3)]


@pytest.mark.parametrize("kernel", ["s", "p", "auto"])      # the product kernels (A/B variants live in the tools build only)
@pytest.mark.parametrize("fp16c", [False, True])
@pytest.mark.parametrize("size", SIZES)
def test_stream_collide_matches_oracle(luw, kernel, fp16c, size):
    from oracle import oracle
    Nx, Ny, Nz = size
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 1e-4, fp16c, kernel, synthetic_state(Nx, Ny, Nz, seed=3, shell="luw"), every_step=True)
    g.run(0); o.initialize()
    check(g, o, "after initialize")
    for _ in range(4):
        g.run(3); o.run(3)          # odd count: exercises both parities across calls
        check(g, o, "t=%d" % o.t)


@pytest.mark.parametrize("kernel,fp16c,Nx", [("s", False, 22), ("s", False, 23), ("s", True, 22), ("p", True, 22), ("p", True, 23), ("s", True, 23)])
def test_periodic_box_without_boundaries(luw, kernel, fp16c, Nx):
    # "all box sides where no boundary type is set remain periodic" (DOCUMENTATION.md:195-256): wrap in x,y,z
    # (pair kernel: the row-end lane reads and writes its wrapped x+1 neighbours in two halves; with an odd Nx the last cell
    # of a row shares its lane with the row padding)
    from oracle import oracle
    Ny, Nz = 9, 7
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 0.02, fp16c, kernel, synthetic_state(Nx, Ny, Nz, seed=5, solids=True, shell=None), every_step=True)
    g.run(9); o.run(9)
    check(g, o, "periodic")


@pytest.mark.parametrize("kernel,fp16c,N,G", [("s", False, (300, 16, 8), 2), ("s", False, (40, 24, 5), 1), ("p", True, (640, 16, 6), 1),
    ("s", True, (70, 8, 9), 1), ("s", False, (300, 12, 8), 1), ("s", False, (130, 70, 4), 2), ("p", True, (256, 44, 4), 4), ("s", False, (520, 33, 3), 4)])
def test_rows_of_a_block_row_on_one_xcd_same_values(luw, kernel, fp16c, N, G):
    """LUW_XCD_ROWS=G (luw_create sets 4 by itself for lattices with large DDF planes): the step kernels remap their workgroups so that the blocks of one
    lattice row run on one XCD, G rows per XCD and turn (xcd_row_order, csrc/luw_device.hpp) -- a permutation of the blocks of the launch's first rows, as
    many as
    make whole turns of 8 G rows; the rows behind them keep the dispatch order (12, 70, 44, 33 rows here).  Same bits as the oracle, several blocks per
    row included"""
    import os
    from latticeurbanwind_amd import capi
    from oracle import oracle
    saved = os.environ.get("LUW_XCD_ROWS")
    os.environ["LUW_XCD_ROWS"] = str(G); capi.reload_tuning()
    try:
        assert "LUW_XCD_ROWS=%d" % G in capi.tuning_text()
        g, o = make_pair(luw, oracle, *N, 0.02, fp16c, kernel, synthetic_state(*N, seed=7, solids=True, shell=None), every_step=True)
        g.run(7); o.run(7)
        check(g, o, "xcd rows")
    finally:
        if saved is None: os.environ.pop("LUW_XCD_ROWS", None)
        else: os.environ["LUW_XCD_ROWS"] = saved
        capi.reload_tuning()


@pytest.mark.parametrize("kernel", ["s", "p"])
@pytest.mark.parametrize("fp16c", [False, True])
def test_all_force_terms(luw, kernel, fp16c):
    # volume force + Coriolis + per-cell force field + buffer nudging (west/south/north/top, east = downstream) + top sponge
    from oracle import oracle
    Nx, Ny, Nz = 40, 36, 30
    nud = dict(n_cells=5, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1)
    spg = dict(n_cells=6, inv_tau=0.02)
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 2e-5, fp16c, kernel, synthetic_state(Nx, Ny, Nz, seed=11, shell="luw"),
                     force=(1e-5, -2e-5, 3e-6), coriolis=(0.0, 3e-5, 4e-5), nudging=nud, sponge=spg, use_F=True, every_step=True)
    for _ in range(3):
        g.run(5); o.run(5)
        check(g, o, "forces t=%d" % o.t)


@pytest.mark.parametrize("forces", ["zones", "zones+coriolis", "coriolis", "none"])
def test_fp16c_pair_kernel_force_modes_match_oracle(luw, forces):
    """FP16C, automatic kernel choice, a row wide enough for the pair kernel: the launch takes the instantiation that fits what can push
    the cells of its box -- zones: everything, switched per wave; Coriolis only: uniform-force mode; nothing: no force path, 5 waves per
    SIMD, TYPE_E lanes relaxed with w = 1.  Every case equals the oracle bit for bit."""
    from oracle import oracle
    Nx, Ny, Nz = 648, 28, 26
    nud = dict(n_cells=5, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1) if "zones" in forces else None
    spg = dict(n_cells=4, inv_tau=0.02) if "zones" in forces else None
    cor = (0.0, 3e-5, 4e-5) if "coriolis" in forces else None
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 2e-5, True, "auto", synthetic_state(Nx, Ny, Nz, seed=21, shell="luw"), coriolis=cor, nudging=nud, sponge=spg,
        every_step=False)
    for _ in range(2):
        g.run(5); o.run(5)
        check(g, o, "force mode %s t=%d" % (forces, o.t))


@pytest.mark.parametrize("grow", ["", "west", "south", "north", "top"])
@pytest.mark.parametrize("coriolis", [False, True])
def test_fp16c_zone_free_core_as_its_own_launch_box(luw, coriolis, grow):
    """A step assembled from launch boxes (the way the multi-domain hosts do it), cut exactly along the force zones: the zone-free core goes out
    as one box -- for which the library picks the instantiation without the force path (or with the uniform forces only) -- and six slabs around it
    take the general kernels.  grow = a face: the same with the middle box reaching two layers INTO that face's zone (the outermost layer of a
    zone carries weight 0) -- the library must then keep the general kernel for it; a wrong idea of the host about where that zone ends
    would show as missing forces in those layers."""
    from oracle import oracle
    Nx, Ny, Nz = 648, 30, 28
    # zones: west 0..5, south 0..5, north Ny-6.., top Nz-6..; east is downstream
    nud = dict(n_cells=5, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1)
    spg = dict(n_cells=4, inv_tau=0.02)                                                 # sponge: the 4 layers under the top one
    cor = (0.0, 3e-5, 4e-5) if coriolis else None
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 2e-5, True, "auto", synthetic_state(Nx, Ny, Nz, seed=22, shell="luw"), coriolis=cor, nudging=nud, sponge=spg,
        every_step=True)
    # first cells behind / before the zones (x stays even, pairs stay whole)
    x0, y0, y1, z1 = 6, 6, Ny - 6, Nz - 6
    if grow == "west": x0 -= 2
    if grow == "south": y0 -= 2
    if grow == "north": y1 += 2
    if grow == "top": z1 += 2
    core = (x0, Nx, y0, y1, 0, z1)
    slabs = [(0, Nx, 0, Ny, z1, Nz), (0, Nx, 0, y0, 0, z1), (0, Nx, y1, Ny, 0, z1), (0, x0, y0, y1, 0, z1)]
    g.run(0)
    for step in range(1, 9):
        for b in [core] + slabs:
            g.enqueue_stream_collide(b, write_fields=True)
        g.increment_time_step(1)
        if step % 4 == 0:
            g.finish(); o.run(4)
            check(g, o, "core box, coriolis %s, grown %s, t=%d" % (coriolis, grow or "nowhere", o.t))


@pytest.mark.parametrize("kernel", ["s", "p"])
def test_fluid_reference_cells_switch_to_fields_every_step(luw, kernel):
    """buffer nudging / top sponge read u of reference cells on the outer faces (FX/kernel.cpp:1543-1611).  With TYPE_E / solid faces (every LUW deck) that u
    is an input and the library writes rho,u in the last step of a run() call only: equal to the oracle whatever the call length.  On a lattice whose faces
    are FLUID (fully periodic here) luw_initialize switches to writing the fields in every step like the reference's UPDATE_FIELDS build -- there the reference
    itself reads a neighbour's u while that neighbour's thread rewrites it in the same launch, so no bit-level expectation exists; the run must be sane"""
    from oracle import oracle
    Nx, Ny, Nz = (512, 10, 12) if kernel == "p" else (40, 28, 24)
    nud = dict(n_cells=3, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1); spg = dict(n_cells=3, inv_tau=0.02)
    for shell, expect in (("luw", False), (None, True)):
        g, o = make_pair(luw, oracle, Nx, Ny, Nz, 2e-5, kernel == "p", kernel, synthetic_state(Nx, Ny, Nz, seed=31, shell=shell), nudging=nud, sponge=spg,
            every_step=False)
        g.run(0)
        assert g.fields_every_step() == expect
        g.run(7); o.run(7)
        if shell:
            check(g, o, "nudging / sponge with input faces, one 7-step call")
        else:
            g.u.read_from_device(); g.rho.read_from_device()
            assert np.isfinite(g.u.data).all() and float(np.abs(g.u.data - o.u).max()) < 2e-2 and float(np.abs(g.rho.data - o.rho).max()) < 2e-2
        g.close()


@pytest.mark.parametrize("kernel", ["s", "p"])
def test_deferred_field_update_equals_every_step(luw, kernel):
    # default mode writes rho,u only in the last step of a run() call; observed values must equal UPDATE_FIELDS
    from oracle import oracle
    Nx, Ny, Nz = 24, 20, 16
    st = synthetic_state(Nx, Ny, Nz, seed=2, shell="luw")
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 1e-4, kernel == "p", kernel, st, every_step=False)
    g.run(7); o.run(7)
    check(g, o, "deferred 7")
    g.run(6); o.run(6)
    check(g, o, "deferred 13")


@pytest.mark.parametrize("size", [(24, 20, 16), (37, 9, 5)])
def test_on_device_time_averaging_matches_host_welford(luw, size):
    # device Welford == the reference's host accumulate_from_buffers (FX/setup.cpp:4441-4488), bit for bit
    from oracle import oracle
    Nx, Ny, Nz = size
    st = synthetic_state(Nx, Ny, Nz, seed=12, shell="luw")
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 1e-3, False, "s", st)
    stats = oracle.OracleStats(o.N)
    g.run(5); o.run(5)
    g.stats_reset()
    for _ in range(6):
        g.run(1); o.run(1)                  # sampled steps: one step per call writes rho,u
        g.stats_accumulate(); stats.accumulate(o)
    d = g.stats_download()
    assert d["count"] == 6
    for k in ("avg_u", "avg_rho", "m2_u", "m2_v", "m2_w"):
        assert np.array_equal(d[k], getattr(stats, k)), k
    # sampling after a multi-step run() is allowed (its last step wrote the fields); stale fields are refused
    g.enqueue_stream_collide((0, Nx, 0, Ny, 0, Nz), False); g.finish()
    with pytest.raises(luw.LuwError):
        g.stats_accumulate()


@pytest.mark.parametrize("size,fp16c,kernel", [((24, 20, 16), False, "s"), ((37, 9, 5), False, "s"), ((40, 12, 6), True, "s"),
                                               ((512, 6, 5), True, "p"), ((259, 5, 4), True, "p")])
@pytest.mark.parametrize("first,stride", [(1, 1), (3, 2)])
def test_sampled_run_with_fused_statistics_matches_host_welford(luw, size, fp16c, kernel, first, stride):
    # luw_run_sampled: sampled steps carry the Welford update in the step kernel's epilogue (scalar and pair kernels, solids,
    # TYPE_E cells, odd row widths).  Must equal the reference's host loop (FX/setup.cpp:4441-4488) on the oracle's fields bit
    # for bit, leave the same rho,u as a plain run, and equal the separate-kernel path (LUW_TEST_AIDS=separate_stats is process-wide, so
    # that path is compared through { run(1); stats_accumulate() }).
    from oracle import oracle
    Nx, Ny, Nz = size
    st = synthetic_state(Nx, Ny, Nz, seed=21, shell="luw")
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 1e-3, fp16c, kernel, st)
    g2, _ = make_pair(luw, oracle, Nx, Ny, Nz, 1e-3, fp16c, kernel, st)
    stats = oracle.OracleStats(o.N)
    g.run(4); o.run(4); g2.run(4)
    g.stats_reset(); g2.stats_reset()
    steps = 9
    g.run_sampled(steps, first, stride)
    n = 0
    for i in range(1, steps + 1):
        o.run(1); g2.run(1)
        if i >= first and (i - first) % stride == 0:
            stats.accumulate(o); g2.stats_accumulate(); n += 1
    check(g, o, "fields after the sampled run")
    d, d2 = g.stats_download(), g2.stats_download()
    assert d["count"] == n == d2["count"]
    for k in ("avg_u", "avg_rho", "m2_u", "m2_v", "m2_w"):
        assert np.array_equal(d[k], getattr(stats, k)), k
        assert np.array_equal(d[k], d2[k]), k


def test_mass_conservation_at_256cubed(luw):
    # size-independent property at a mid size: the LUW shell state stays finite and the fluid mass away from the inflow cells is kept
    from latticeurbanwind_amd import capi
    N = 256
    st = synthetic_state(N, N, N, seed=9, shell="luw")
    g = luw.LBM(N, N, N, 1e-5, kernel=capi.KERNEL_SCALAR)
    g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
    g.run(10)
    g.u.read_from_device(); g.rho.read_from_device()
    fluid = (st[0] & TYPE_S) == 0
    sel = fluid & ((st[0] & TYPE_E) == 0)
    m0 = st[2][sel].astype(np.float64).sum()
    assert np.isfinite(g.u.data).all() and abs(g.rho.data[sel].astype(np.float64).sum() / m0 - 1) < 1e-3
    g.close()


def test_fp16c_kernels_agree_at_bench_class_size(luw):
    """FP16C at 512x256x256 with the LUW shell: the pair kernel (the automatic choice at this width) and the scalar kernel leave
    the same bits in u, rho and every DDF plane"""
    from latticeurbanwind_amd import capi
    N = (512, 256, 256)
    st = synthetic_state(*N, seed=10, shell="luw")
    res = []
    for k in (capi.KERNEL_AUTO, capi.KERNEL_SCALAR):
        g = luw.LBM(*N, 1e-5, fp16c=True, kernel=k)
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
        g.run(11)
        g.u.read_from_device(); g.rho.read_from_device()
        res.append((g.u.data.copy(), g.rho.data.copy(), np.asarray(g.download_fi()).copy()))
        g.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    assert np.isfinite(res[0][0]).all()


@pytest.mark.parametrize("fp16c", [False, True])
def test_wide_lattice_matches_oracle(luw, fp16c):
    """384x96x64 (2.4 M cells, rows wide enough for the pair kernel and for three 128-lane blocks per row), 12 steps, forces on:
    the product kernels of both formats against the oracle, value for value"""
    from oracle import oracle
    Nx, Ny, Nz = 384, 96, 64
    nud = dict(n_cells=6, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1)
    g, o = make_pair(luw, oracle, Nx, Ny, Nz, 2e-5, fp16c, "auto", synthetic_state(Nx, Ny, Nz, seed=13, shell="luw"),
                     coriolis=(0.0, 3e-5, 4e-5), nudging=nud, every_step=False)
    g.run(12); o.run(12)
    check(g, o, "wide lattice t=12")


def test_rest_state_and_mass_conservation_at_512cubed(luw):
    # BASELINE config C2 size (512^3 FP32): rest state is an exact fixed point; a periodic box conserves mass
    N = 512
    g = luw.LBM(N, N, N, 1e-5)
    g.run(4)
    g.rho.read_from_device(); g.u.read_from_device()
    assert np.all(g.rho.data == 1.0) and not g.u.data.any()
    g.close()
    g = luw.LBM(N, N, N, 0.01)
    x = np.arange(N, dtype=np.float32)
    wave = (0.02 * np.sin(2 * np.pi * x / N)).astype(np.float32)
    g.u.data.reshape(3, N, N, N)[0] = wave[None, :, None]       # u_x varies along y: shear wave, divergence free
    g.run(20)
    g.rho.read_from_device()
    assert abs(g.rho.data.astype(np.float64).mean() - 1.0) < 1e-7
    g.close()


def test_building_array_at_1024x1024x256(luw):
    """BASELINE configs[2] size and solid mask (bench.py --buildings): with solids, the rest state stays an exact fixed point
    (bounce-back moves only zeros), a periodic box with the building array conserves the fluid mass while a shear wave
    decays."""
    import sys
    sys.path.insert(0, ROOT)
    from bench import channel_state
    Nx, Ny, Nz = 1024, 1024, 256
    fl, u, rho = channel_state(Nx, Ny, Nz, buildings=True)
    solid = (fl & TYPE_S) != 0
    g = luw.LBM(Nx, Ny, Nz, 1e-5)
    g.flags.data[:] = np.where(solid, TYPE_S, 0).astype(np.uint8)        # periodic box: solids only
    g.run(6)
    g.rho.read_from_device(); g.u.read_from_device()
    assert np.all(g.rho.data == 1.0) and not g.u.data.any()
    g.close()
    g = luw.LBM(Nx, Ny, Nz, 0.01)
    g.flags.data[:] = np.where(solid, TYPE_S, 0).astype(np.uint8)
    wave = (0.02 * np.sin(2 * np.pi * np.arange(Nz, dtype=np.float32) / Nz)).astype(np.float32)
    ux = g.u.data.reshape(3, Nz, Ny, Nx)[0]
    ux[:] = wave[:, None, None]; ux[solid.reshape(Nz, Ny, Nx)] = 0.0
    g.run(20)
    g.rho.read_from_device(); g.u.read_from_device()
    fluid = ~solid
    assert np.isfinite(g.u.data).all()
    assert abs(g.rho.data[fluid].astype(np.float64).mean() - 1.0) < 1e-6       # bounce-back conserves mass
    assert not g.u.data.reshape(3, -1)[:, solid].any()                        # solids never receive a velocity
    g.close()


@pytest.mark.parametrize("case,fp16c,npz", [("CaseA", False, "ref_fp32_CaseA.npz"), ("CaseB", False, "ref_fp32_CaseB.npz"),
                                            ("CaseL", False, "ref_fp32_CaseL.npz"), ("CaseA", True, "ref_shipped_CaseA.npz"),
                                            ("CaseL", True, "ref_shipped_CaseL.npz")])
def test_hip_path_vs_real_reference_fields(luw, case, fp16c, npz):
    # the committed fields of the REAL reference solver (FluidX3D via OpenCL on MI355X): u at K = 8 and K = 64 and the mean over the last four steps (u_avg),
    # each under its ceiling and within twice the recorded observation (helpers.check_gate; the values are the oracle's, bit for bit)
    from oracle import setup_profile
    from test_oracle_vs_reference import run_and_compare
    gold = np.load(os.path.join(GOLD, npz))
    s = setup_profile.setup_profile_case(os.path.join(GOLD, "refcases", case, "conf.luwpf"), solid_mask=gold["solid"])
    nud = dict(n_cells=s["buffer_N"], inv_tau=float(s["buffer_inv_tau"]), downstream_face=s["buffer_face"], nudge_vertical=s["buffer_nudge_vertical"]) if s[
        "buffer_active"] else None
    spg = dict(n_cells=s["sponge_N"], inv_tau=float(s["sponge_inv_tau"])) if s["sponge_active"] else None
    g = luw.LBM(s["Nx"], s["Ny"], s["Nz"], float(s["nu"]), fp16c=fp16c, buffer_nudging=nud, top_sponge=spg)
    g.flags.data[:] = s["flags"]; g.u.data[:] = s["u"]; g.rho.data[:] = s["rho"]

    def dev_u(l):
        l.u.read_from_device(); return l.u.data

    def dev_rho(l):
        l.rho.read_from_device(); return l.rho.data
    run_and_compare(g, gold, s, "oracle:" + npz[:-4], 2e-6 if fp16c else 2e-7, 6e-5 if fp16c else 1e-6, dev_u, dev_rho)
    g.close()


def test_halo_extract_insert_match_oracle(luw):
    # transfer_extract_fi / transfer__insert_fi (FX/kernel.cpp:2241-2270) on a domain that is split in all 3 axes
    import ctypes as C
    import torch
    from oracle import oracle
    Nx, Ny, Nz = 14, 10, 9
    st = synthetic_state(Nx, Ny, Nz, seed=4, shell=None)
    for fp16c in (False, True):
        g = luw.LBM(Nx, Ny, Nz, 0.01, fp16c=fp16c, D=(2, 2, 2), O=(-1, -1, -1), update_fields_every_step=True)
        o = oracle.OracleLBM(Nx, Ny, Nz, 0.01, fp16c=fp16c, D=(2, 2, 2), O=(-1, -1, -1))
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
        o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
        g.run(0); o.initialize()
        for step in range(3):
            g.enqueue_stream_collide((0, Nx, 0, Ny, 0, Nz), True); g.finish()
            o.stream_collide()
            tdt = torch.int16 if fp16c else torch.float32
            for d in range(3):
                A = g.area(d)
                bp = torch.zeros(5 * A, dtype=tdt, device="cuda"); bm = torch.zeros(5 * A, dtype=tdt, device="cuda")
                torch.cuda.synchronize()
                g.enqueue_extract_fi(d, bp.data_ptr(), bm.data_ptr()); g.finish()
                obp, obm = o.extract_fi(d)
                # same elements; inside a y face the library runs x fastest (a = x + z Nx) where the reference has a = z + x Nz
                order = (lambda b: b.reshape(5, Nx, Nz).transpose(0, 2, 1).ravel()) if d == 1 else (lambda b: b)
                assert np.array_equal(bp.cpu().numpy().view(obp.dtype), order(obp)) and np.array_equal(bm.cpu().numpy().view(obm.dtype), order(obm))
                # swap p/m (periodic self-neighbour) and insert
                g.enqueue_insert_fi(d, bm.data_ptr(), bp.data_ptr()); g.finish()
                o.insert_fi(d, obm, obp)
            g.increment_time_step(1); o.t += 1
            assert ddf_equal(g.download_fi(), o.fi)
        g.close()


@pytest.mark.parametrize("fp16c", [False, True])
@pytest.mark.parametrize("sponge", [False, True])
def test_thermal_lattice_matches_oracle(luw, fp16c, sponge):
    """TEMPERATURE (D3Q7 advection-diffusion of T on the LBM velocity, TYPE_T presets, top sponge on T; FX/kernel.cpp:1306-1335,
    1639-1684): T, the thermal DDFs and -- unchanged by it -- u, rho and the DDFs equal the oracle value for value"""
    from oracle import oracle
    Nx, Ny, Nz = 40, 28, 24
    st = synthetic_state(Nx, Ny, Nz, seed=11, shell="luw")
    spg = dict(n_cells=5, inv_tau=0.02) if sponge else None
    g = luw.LBM(Nx, Ny, Nz, 1e-3, fp16c=fp16c, alpha=4e-3, top_sponge=spg)
    o = oracle.OracleLBM(Nx, Ny, Nz, 1e-3, fp16c=fp16c, alpha=4e-3)
    if sponge:
        o.set_top_sponge(5, 0.02)
    flags = st[0].copy()
    E = (flags & 3) == 2
    flags[E] |= TYPE_T                                          # temperature boundary on the velocity boundary
    rng = np.random.default_rng(3)
    extra = (rng.random(flags.size) < 0.01) & ((flags & 3) == 0)
    flags[extra] |= TYPE_T                                       # a few interior heat sources
    z = (np.arange(flags.size) // (Nx * Ny)).astype(np.float32)
    Tinit = np.ones(flags.size, np.float32)
    Tinit[(flags & TYPE_T) != 0] = (1.0 + 0.05 * np.sin(z / 4.0) + 0.02 * rng.standard_normal(flags.size).astype(np.float32))[(flags & TYPE_T) != 0]
    for l in (g, o):
        (l.flags.data if hasattr(l.flags, "data") else l.flags)[:] = flags
        (l.u.data if hasattr(l.u, "data") else l.u)[:] = st[1]
        (l.rho.data if hasattr(l.rho, "data") else l.rho)[:] = st[2]
        (l.T.data if hasattr(l.T, "data") else l.T)[:] = Tinit
    g.run(0)
    assert not g.fields_every_step()       # every temperature the kernel reads is a preset: T, like rho and u, is stored by the last step of a call only
    for steps in (1, 2, 14):
        g.run(steps); o.run(steps)
        g.T.read_from_device()
        assert np.array_equal(g.T.data, o.T), "T after %d more steps: %d cells differ" % (steps, int((g.T.data != o.T).sum()))
        assert ddf_equal(g.download_gi(), o.gi)
        check(g, o, "flow fields with the thermal lattice on")
    assert np.isfinite(o.T).all() and o.T[(flags & 1) == 0].std() > 1e-4
    g.close()


def test_sponge_on_a_computed_top_temperature_stores_T_every_step(luw):
    """the top sponge pulls T towards the top layer's value (FX/kernel.cpp:1660-1666).  Where that layer is preset (TYPE_T, every LUW deck with
    temperatures) T is an input and is stored by the last step of a call only; a top layer whose temperature is computed has to be stored by
    every step, like the reference does"""
    Nx, Ny, Nz = 40, 28, 24
    st = synthetic_state(Nx, Ny, Nz, seed=12, shell="luw")
    for preset_top, expect in ((True, False), (False, True)):
        g = luw.LBM(Nx, Ny, Nz, 1e-3, alpha=4e-3, top_sponge=dict(n_cells=5, inv_tau=0.02))
        flags = st[0].copy()
        if preset_top:
            flags.reshape(Nz, Ny, Nx)[Nz - 1] |= TYPE_T
        g.flags.data[:] = flags; g.u.data[:] = st[1]; g.rho.data[:] = st[2]; g.T.data[:] = 1.0
        g.run(0)
        assert g.fields_every_step() == expect
        g.run(5); g.T.read_from_device()
        assert np.isfinite(g.T.data).all()
        g.close()


@pytest.mark.parametrize("size", [(514, 5, 6), (257, 4, 5), (640, 6, 4), (130, 7, 5)])
@pytest.mark.parametrize("forces", ["none", "coriolis", "zones"])
def test_thermal_lattice_in_the_pair_kernel(luw, size, forces):
    """FP16C + thermal lattice on rows wide enough for the pair kernel (both lattices two cells per lane, the second set of values parked in
    LDS): even and odd widths (the row-end lane's wrap fix-up and the single-cell tail lane of the +x plane), TYPE_T presets on the TYPE_E shell
    and inside, top sponge on T, all three force instantiations -- T, the thermal DDFs, u, rho and the DDFs against the oracle"""
    from oracle import oracle
    Nx, Ny, Nz = size
    st = synthetic_state(Nx, Ny, Nz, seed=13, shell="luw")
    spg = dict(n_cells=2, inv_tau=0.02)
    nud = dict(n_cells=3, inv_tau=0.0133333, downstream_face=2, nudge_vertical=1) if forces == "zones" else None
    g = luw.LBM(Nx, Ny, Nz, 1e-3, fp16c=True, alpha=4e-3, top_sponge=spg, buffer_nudging=nud)
    o = oracle.OracleLBM(Nx, Ny, Nz, 1e-3, fp16c=True, alpha=4e-3)
    o.set_top_sponge(2, 0.02)
    if nud: o.set_buffer_nudging(3, 0.0133333, 2, 1)
    if forces != "none":
        g.set_coriolis(0.0, 3e-5, 4e-5); o.set_coriolis(0.0, 3e-5, 4e-5)
    flags = st[0].copy()
    flags[(flags & 3) == 2] |= TYPE_T
    rng = np.random.default_rng(5)
    flags[(rng.random(flags.size) < 0.02) & ((flags & 3) == 0)] |= TYPE_T
    Tinit = np.ones(flags.size, np.float32)
    hot = (flags & TYPE_T) != 0
    Tinit[hot] = (1.0 + 0.05 * rng.standard_normal(flags.size).astype(np.float32))[hot]
    for l in (g, o):
        (l.flags.data if hasattr(l.flags, "data") else l.flags)[:] = flags
        (l.u.data if hasattr(l.u, "data") else l.u)[:] = st[1]
        (l.rho.data if hasattr(l.rho, "data") else l.rho)[:] = st[2]
        (l.T.data if hasattr(l.T, "data") else l.T)[:] = Tinit
    for steps in (1, 2, 6):
        g.run(steps); o.run(steps)
        g.T.read_from_device()
        assert np.array_equal(g.T.data, o.T), "%s %s: T after %d more steps: %d cells differ" % (size, forces, steps, int((g.T.data != o.T).sum()))
        assert ddf_equal(g.download_gi(), o.gi), "%s %s: thermal DDFs" % (size, forces)
        check(g, o, "flow fields with the thermal lattice in the pair kernel")
    g.close()


def test_fp32_row_form_addressing_matches_oracle():
    """FP32 lattices whose planes exceed 32-bit byte offsets (beyond 2^30 cells per GPU, e.g. 1024^3) take the row-form addressing of
    the scalar kernel; the oracle cannot run at that size, so the same code path is forced on the small parity cases
    (LUW_TEST_AIDS=addr_row) in a child process: step, forces, thermal lattice and multi-domain halos, FP32."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_parity.py"), os.path.join(root, "tests", "test_gpu_halo.py"),
                        "-k", "(stream_collide_matches_oracle and False-s) or (all_force_terms and False-s) or thermal_lattice or (local_group_equals and "
                            "False)"],
                       env=dict(os.environ, LUW_TEST_AIDS="addr_row"), capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-500:]


def test_1024_cubed_on_one_gpu_properties():
    """the north-star grid on ONE GPU, FP32: 82 GB of DDFs, planes beyond 32-bit byte offsets (row-form addressing at its real
    size).  No oracle at 1.07 G cells: the rest state must be an exact fixed point and a periodic shear wave must conserve mass
    (tools/check_huge.py, in a child process so that its 35 GB of host arrays are gone afterwards)."""
    import subprocess, sys
    import re
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_huge.py"), "f32"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and "exact fixed point = True" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    # (its 2^32-byte planes are the largest the flat addressing form takes: byte offsets up to 2^32 - 4.)  The row form at the same size
    # must leave the same 4.3 G values, bit for bit (digest over rho and u):
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_huge.py"), "f32", "wave"], capture_output=True, text=True, timeout=1500, cwd=ROOT,
        env=dict(os.environ, LUW_TEST_AIDS="addr_row"))
    assert r2.returncode == 0 and "shear wave, 12 steps" in r2.stdout, r2.stdout[-1500:] + r2.stderr[-1500:]
    d1, d2 = re.search(r"digest (xor=\w+ sum=\w+)", r.stdout), re.search(r"digest (xor=\w+ sum=\w+)", r2.stdout)
    assert d1 and d2 and d1.group(1) == d2.group(1), (d1 and d1.group(1), d2 and d2.group(1))


def test_random_force_zones_match_oracle(luw):
    """tests/fuzz/fuzz_zones.py, a fixed draw: nudging / sponge zones of random thickness (thin, thicker than a wave's 128 cells, W / E zones that overlap on
    narrow lattices), random downstream face, odd and even rows, Coriolis / volume force at random -- the general pair kernel with its early reference
    fetch and both one-cell kernels against the oracle, every value."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_zones", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz", "fuzz_zones.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.run(30, 11, say=lambda t: None) == 30
