"""ctypes binding of the CPU oracle (oracle/luw_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (latticeurbanwind_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TYPE_S, TYPE_E, TYPE_T = 0x01, 0x02, 0x04


class Cfg(C.Structure):
    _fields_ = [
        ("Nx", C.c_uint32), ("Ny", C.c_uint32), ("Nz", C.c_uint32),
        ("Dx", C.c_uint32), ("Dy", C.c_uint32), ("Dz", C.c_uint32),
        ("Ox", C.c_int32), ("Oy", C.c_int32), ("Oz", C.c_int32),
        ("w", C.c_float),
        ("fx", C.c_float), ("fy", C.c_float), ("fz", C.c_float),
        ("omega_x", C.c_float), ("omega_y", C.c_float), ("omega_z", C.c_float),
        ("fp16c", C.c_int32), ("subgrid", C.c_int32),
        ("buffer_active", C.c_int32), ("buffer_N", C.c_uint32), ("buffer_inv_tau", C.c_float),
        ("buffer_nudge_vertical", C.c_int32), ("downstream_face", C.c_int32),
        ("sponge_active", C.c_int32), ("sponge_N", C.c_uint32), ("sponge_inv_tau", C.c_float),
        ("w_T", C.c_float),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libluw_oracle.so")
    src = os.path.join(_HERE, "luw_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        cfgp = C.POINTER(Cfg)
        L.luwo_half_to_float_custom.argtypes = [C.c_uint16]; L.luwo_half_to_float_custom.restype = C.c_float
        L.luwo_float_to_half_custom.argtypes = [C.c_float]; L.luwo_float_to_half_custom.restype = C.c_uint16
        L.luwo_literal_roundtrip.argtypes = [C.c_float]; L.luwo_literal_roundtrip.restype = C.c_float
        L.luwo_calculate_f_eq.argtypes = [C.c_float] * 4 + [vp]; L.luwo_calculate_f_eq.restype = None
        L.luwo_calculate_rho_u.argtypes = [vp] * 5; L.luwo_calculate_rho_u.restype = None
        L.luwo_initialize.argtypes = [cfgp, vp, vp, vp, vp]; L.luwo_initialize.restype = None
        L.luwo_stream_collide.argtypes = [cfgp, vp, vp, vp, vp, vp, u64]; L.luwo_stream_collide.restype = None
        L.luwo_run.argtypes = [cfgp, vp, vp, vp, vp, vp, u64, u64]; L.luwo_run.restype = None
        L.luwo_initialize_thermal.argtypes = [cfgp, vp, vp, vp, vp]; L.luwo_initialize_thermal.restype = None
        L.luwo_stream_collide_thermal.argtypes = [cfgp, vp, vp, vp, vp, vp, vp, vp, u64]; L.luwo_stream_collide_thermal.restype = None
        L.luwo_get_area.argtypes = [cfgp, u32]; L.luwo_get_area.restype = u64
        L.luwo_transfer_extract_fi.argtypes = [cfgp, u32, u64, vp, vp, vp]; L.luwo_transfer_extract_fi.restype = None
        L.luwo_transfer_insert_fi.argtypes = [cfgp, u32, u64, vp, vp, vp]; L.luwo_transfer_insert_fi.restype = None
        L.luwo_transfer_extract_gi.argtypes = [cfgp, u32, u64, vp, vp, vp]; L.luwo_transfer_extract_gi.restype = None
        L.luwo_transfer_insert_gi.argtypes = [cfgp, u32, u64, vp, vp, vp]; L.luwo_transfer_insert_gi.restype = None
        L.luwo_moments.argtypes = [cfgp, vp, u64, vp, vp]; L.luwo_moments.restype = None
        L.luwo_vk_inlet_apply.argtypes = [u64, u32, C.c_float, C.c_float, C.c_float, u64, u64, vp, vp, vp, vp, vp]; L.luwo_vk_inlet_apply.restype = None
        L.luwo_accumulate_stats.argtypes = [u64, u64] + [vp] * 7; L.luwo_accumulate_stats.restype = None
        L.luwo_stream_collide_literal.argtypes = [cfgp, vp, vp, vp, vp, vp, vp, vp, u64]; L.luwo_stream_collide_literal.restype = None
        L.luwo_set_fast.argtypes = [C.c_int]; L.luwo_set_fast.restype = None
        L.luwo_get_fast.restype = C.c_int
        L.luwo_set_threads.argtypes = [C.c_int]; L.luwo_set_threads.restype = None
        L.luwo_get_max_threads.restype = C.c_int
        L.luwo_copy_bandwidth_gbps.argtypes = [u64]; L.luwo_copy_bandwidth_gbps.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def w_from_nu(nu):
    """def_w as the reference kernel sees it: 1/(3 nu + 1/2) in FP32 (FX/lbm.hpp:140), printed with
    to_string(float) and parsed back (FX/lbm.cpp:664)."""
    tau = np.float32(3.0) * np.float32(nu) + np.float32(0.5)
    return float(lib().luwo_literal_roundtrip(float(np.float32(1.0) / tau)))


def literal(x):
    return float(lib().luwo_literal_roundtrip(float(np.float32(x))))


class OracleLBM:
    """One LBM_Domain of the reference (FX/lbm.hpp:26-219) on the CPU: owns fi/rho/u/flags/F host arrays in
    the reference's layout (rho[N], u[3N] SoA, flags[N], fi[19N] SoA), n = x+(y+z*Ny)*Nx."""

    def __init__(self, Nx, Ny, Nz, nu, fx=0.0, fy=0.0, fz=0.0, fp16c=False, D=(1, 1, 1), O=(0, 0, 0),
                 subgrid=True, use_F=False, alpha=None):
        self.cfg = Cfg()
        c = self.cfg
        c.Nx, c.Ny, c.Nz = Nx, Ny, Nz
        c.Dx, c.Dy, c.Dz = D
        c.Ox, c.Oy, c.Oz = O
        c.w = w_from_nu(nu)
        c.fx, c.fy, c.fz = fx, fy, fz
        c.fp16c = int(bool(fp16c))
        c.subgrid = int(bool(subgrid))
        self.N = Nx * Ny * Nz
        self.rho = np.ones(self.N, np.float32)
        self.u = np.zeros(3 * self.N, np.float32)
        self.flags = np.zeros(self.N, np.uint8)
        self.F = np.zeros(3 * self.N, np.float32) if use_F else None
        self.fi = np.zeros(19 * self.N, np.uint16 if fp16c else np.float32)
        # TEMPERATURE extension (FX/lbm.hpp:140-141, FX/lbm.cpp:750): thermal diffusivity alpha -> def_w_T through the text round trip
        self.thermal = alpha is not None
        if self.thermal:
            c.w_T = literal(np.float32(1.0) / (np.float32(2.0) * np.float32(alpha) + np.float32(0.5)))
            self.T = np.ones(self.N, np.float32)
            self.gi = np.zeros(7 * self.N, np.uint16 if fp16c else np.float32)
        self.t = 0
        self.initialized = False

    # FX/lbm.hpp:496, FX/setup.cpp:3844-3903 (process-global config consumed by device_defines)
    def set_coriolis(self, ox, oy, oz):
        self.cfg.omega_x, self.cfg.omega_y, self.cfg.omega_z = ox, oy, oz

    def set_buffer_nudging(self, n_cells, inv_tau, downstream_face, nudge_vertical=0):
        c = self.cfg
        c.buffer_active, c.buffer_N, c.buffer_inv_tau = 1, int(n_cells), literal(inv_tau)
        c.downstream_face, c.buffer_nudge_vertical = int(downstream_face), int(nudge_vertical)

    def set_top_sponge(self, n_cells, inv_tau):
        c = self.cfg
        c.sponge_active, c.sponge_N, c.sponge_inv_tau = 1, int(n_cells), literal(inv_tau)

    def initialize(self):
        lib().luwo_initialize(C.byref(self.cfg), _p(self.fi), _p(self.rho), _p(self.u), _p(self.flags))
        if self.thermal:
            lib().luwo_initialize_thermal(C.byref(self.cfg), _p(self.gi), _p(self.T), _p(self.u), _p(self.flags))
        self.initialized = True
        self.t = 0

    def run(self, steps):
        if not self.initialized:
            self.initialize()
        if self.thermal:
            for k in range(steps):
                lib().luwo_stream_collide_thermal(C.byref(self.cfg), _p(self.fi), _p(self.rho), _p(self.u), _p(self.flags), _p(self.F),
                                                  _p(self.gi), _p(self.T), self.t + k)
        else:
            lib().luwo_run(C.byref(self.cfg), _p(self.fi), _p(self.rho), _p(self.u), _p(self.flags), _p(self.F),
                           self.t, steps)
        self.t += steps

    def stream_collide(self):
        """one kernel launch at the current t WITHOUT incrementing t (multi-domain drivers do the halo
        exchange between this and increment, FX/lbm.cpp:1262-1290)"""
        lib().luwo_stream_collide(C.byref(self.cfg), _p(self.fi), _p(self.rho), _p(self.u), _p(self.flags),
                                  _p(self.F), self.t)

    def area(self, direction):
        return int(lib().luwo_get_area(C.byref(self.cfg), direction))

    def extract_fi(self, direction, t=None):
        A = self.area(direction)
        bp = np.zeros(5 * A, self.fi.dtype); bm = np.zeros(5 * A, self.fi.dtype)
        lib().luwo_transfer_extract_fi(C.byref(self.cfg), direction, self.t if t is None else t, _p(bp), _p(bm), _p(self.fi))
        return bp, bm

    def insert_fi(self, direction, bp, bm, t=None):
        lib().luwo_transfer_insert_fi(C.byref(self.cfg), direction, self.t if t is None else t, _p(bp), _p(bm), _p(self.fi))

    def extract_gi(self, direction, t=None):
        A = self.area(direction)
        bp = np.zeros(A, self.gi.dtype); bm = np.zeros(A, self.gi.dtype)
        lib().luwo_transfer_extract_gi(C.byref(self.cfg), direction, self.t if t is None else t, _p(bp), _p(bm), _p(self.gi))
        return bp, bm

    def insert_gi(self, direction, bp, bm, t=None):
        lib().luwo_transfer_insert_gi(C.byref(self.cfg), direction, self.t if t is None else t, _p(bp), _p(bm), _p(self.gi))

    def stream_collide_thermal(self):
        lib().luwo_stream_collide_thermal(C.byref(self.cfg), _p(self.fi), _p(self.rho), _p(self.u), _p(self.flags), _p(self.F), _p(self.gi), _p(self.T), self.t)

    def moments(self):
        rho = np.zeros(self.N, np.float32); u = np.zeros(3 * self.N, np.float32)
        lib().luwo_moments(C.byref(self.cfg), _p(self.fi), self.t, _p(rho), _p(u))
        return rho, u


def vk_inlet_apply(o, tables, use_interp, t0, t1, alpha):
    """kernel vk_inlet_apply (FX/kernel.cpp:2495-2571) on an OracleLBM's u field"""
    lib().luwo_vk_inlet_apply(o.N, int(use_interp), float(t0), float(t1), float(alpha), tables["point_count"], tables["mode_count"],
                              _p(tables["point_cell"]), _p(tables["point_face"]), _p(tables["point_data"]), _p(tables["mode_data"]), _p(o.u))


class OracleStats:
    """host time averaging of the reference's run loop (FX/setup.cpp:4252-4268,4441-4488)"""

    def __init__(self, N):
        self.N = N
        self.avg_u = np.zeros(3 * N, np.float32); self.avg_rho = np.zeros(N, np.float32)
        self.m2_u = np.zeros(N, np.float32); self.m2_v = np.zeros(N, np.float32); self.m2_w = np.zeros(N, np.float32)
        self.count = 0

    def accumulate(self, o):
        self.count += 1
        lib().luwo_accumulate_stats(self.N, self.count, _p(o.rho), _p(o.u), _p(self.avg_u), _p(self.avg_rho), _p(self.m2_u), _p(self.m2_v), _p(self.m2_w))


def set_threads(n):
    lib().luwo_set_threads(int(n))


def copy_bandwidth_gbps(nbytes):
    """host copy bandwidth with the current OpenMP thread count (read + write bytes per second, GB/s)"""
    return float(lib().luwo_copy_bandwidth_gbps(int(nbytes)))


def max_threads():
    return int(lib().luwo_get_max_threads())


def feq(rho, ux, uy, uz):
    out = np.zeros(19, np.float32)
    lib().luwo_calculate_f_eq(rho, ux, uy, uz, _p(out))
    return out


def half_to_float(codes):
    L = lib()
    return np.array([L.luwo_half_to_float_custom(int(c)) for c in np.asarray(codes).ravel()], np.float32)


def float_to_half(vals):
    L = lib()
    return np.array([L.luwo_float_to_half_custom(float(v)) for v in np.asarray(vals, np.float32).ravel()], np.uint16)


def set_fast(on):
    """row-wise path of luw_oracle.c on (default) / off (every call takes the literal one-cell-at-a-time path)"""
    lib().luwo_set_fast(int(bool(on)))


def fast_available():
    return bool(lib().luwo_get_fast())
