"""Error behaviour of the C-ABI on a GPU box: what the reference answers with print_error + exit (FX/lbm.cpp:1123-1142,
FX/utilities.hpp:4370-4382) comes back as a negative status and a message in luw_last_error(); nothing falls back to a CPU path."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(capi, **kw):
    c = capi.Config(); c.struct_size = C.sizeof(capi.Config)
    c.Nx, c.Ny, c.Nz = 8, 8, 8; c.Dx = c.Dy = c.Dz = 1; c.nu = 0.01
    for k, v in kw.items(): setattr(c, k, v)
    return c


def _create(L, cfg):
    h = C.c_void_p()
    return L.luw_create(C.byref(cfg), C.byref(h)), h


@pytest.mark.parametrize("kw,msg", [
    (dict(Nx=0), "Grid point number is 0"), (dict(Dx=0), "0 LBM grid domains"), (dict(nu=0.0), "Viscosity cannot be 0"),
        (dict(nu=-1.0), "Viscosity cannot be negative"),
    (dict(ddf_format=7), "unknown ddf_format"), (dict(struct_size=12), "size mismatch"), (dict(Dx=2, Nx=2), "split axes need"),
        (dict(device=99), "no such HIP device"),
    (dict(buffer_nudging_active=1, buffer_n_cells=0), "buffer_n_cells"), (dict(options=8, alpha=-0.5), "thermal diffusivity"),
    (dict(Nx=2048, Ny=2048, Nz=1024), "2^32")])      # 4.3 G cells: beyond the 32-bit cell index (refused before anything is allocated)
def test_create_rejects_bad_configurations(luw, kw, msg):
    from latticeurbanwind_amd import capi
    L = capi.load()
    rc, h = _create(L, _cfg(capi, **kw))
    assert rc < 0 and not h and msg in L.luw_last_error().decode()


def test_call_order_and_argument_checks(luw):
    from latticeurbanwind_amd import capi
    L = capi.load()
    rc, h = _create(L, _cfg(capi))
    assert rc == 0 and h
    buf = np.zeros(8 * 8 * 8 * 7, np.float32)
    assert L.luw_enqueue_stream_collide(h, 0, 8, 0, 8, 0, 8, 0) == capi.ERR_STATE and "luw_initialize first" in L.luw_last_error().decode()
    assert L.luw_download_gi(h, buf.ctypes.data_as(C.c_void_p)) == capi.ERR_STATE and "LUW_OPT_TEMPERATURE" in L.luw_last_error().decode()
    assert L.luw_stats_accumulate(h) == capi.ERR_STATE and "luw_stats_reset first" in L.luw_last_error().decode()
    assert L.luw_run_sampled(h, 4, 1, 1) == capi.ERR_STATE and "luw_stats_reset first" in L.luw_last_error().decode()
    fused = C.c_int32(7)
    assert L.luw_stats_begin_sample(h, C.byref(fused)) == capi.ERR_STATE and "luw_stats_reset first" in L.luw_last_error().decode()
    assert L.luw_run(h, 3) == 0 and L.luw_get_t(h) == 3                                   # run() initialises on first use, FX/lbm.cpp:1294-1296
    # LUW_WF_SAMPLE without statistics
    assert L.luw_enqueue_stream_collide(h, 0, 8, 0, 8, 0, 8, 2) == capi.ERR_STATE and "luw_stats_begin_sample" in L.luw_last_error().decode()
    assert L.luw_stats_reset(h) == 0
    assert L.luw_run_sampled(h, 4, 0, 1) == capi.ERR_INVALID and "count from 1" in L.luw_last_error().decode()
    assert L.luw_enqueue_stream_collide(h, 0, 8, 0, 8, 0, 8, 2) == capi.ERR_STATE          # no sample counted yet
    assert L.luw_stats_begin_sample(h, C.byref(fused)) == 0 and fused.value == 1
    assert L.luw_enqueue_stream_collide(h, 0, 8, 0, 8, 0, 8, 2) == 0 and L.luw_finish(h) == 0 and L.luw_increment_time_step(h, 1) == 0
    assert L.luw_run(h, 2) == 0 and L.luw_get_t(h) == 6
    L.luw_reset_time_step(h); assert L.luw_run(h, 3) == 0 and L.luw_get_t(h) == 3
    assert L.luw_enqueue_stream_collide(h, 0, 9, 0, 8, 0, 8, 0) == capi.ERR_INVALID and "exceeds the local lattice" in L.luw_last_error().decode()
    assert L.luw_enqueue_stream_collide(h, 4, 4, 0, 8, 0, 8, 0) == 0                      # empty box: nothing to do
    cells = np.array([8 * 8 * 8], np.uint64)
    assert L.luw_gather_attach(h, 1, cells.ctypes.data_as(C.c_void_p)) == capi.ERR_INVALID and "outside the lattice" in L.luw_last_error().decode()
    assert L.luw_upload(None, 1) < 0 and L.luw_run(None, 1) < 0
    L.luw_destroy(h); L.luw_destroy(None)                                                  # destroy(NULL) is a no-op


def test_two_dimensional_and_single_cell_lattices(luw):
    """degenerate extents (Nz = 1 planes, a single row, one cell): every axis wraps onto itself"""
    from oracle import oracle
    from helpers import synthetic_state
    for N in ((16, 12, 1), (20, 1, 1), (1, 1, 1), (1, 9, 1)):
        st = synthetic_state(*N, seed=4, shell=None)
        g = luw.LBM(*N, 0.02); o = oracle.OracleLBM(*N, 0.02)
        g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
        o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
        g.run(5); o.run(5)
        g.u.read_from_device(); g.rho.read_from_device()
        assert np.array_equal(g.u.data, o.u) and np.array_equal(g.rho.data, o.rho) and np.array_equal(g.download_fi(), o.fi), N
        g.close()
