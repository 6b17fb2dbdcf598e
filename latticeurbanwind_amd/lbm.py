"""Host-side mirror of the reference's `LBM` class (FX/lbm.hpp:223-633) over the C-ABI.

Same member names and argument meaning as the reference so that set-up code and tests read like the reference's
own driver code (FX/setup.cpp:6018-6078): construct, fill `flags`, `u.x/y/z`, `rho` through the global index
n = x + (y + z*Ny)*Nx, `run(0)` to upload + initialise, `run(steps)`, `u.read_from_device()`.
Single domain per object (one process per GPU); multi-GPU runs compose these through distributed.py.
"""
import ctypes as C

import numpy as np

from . import capi


class _Vec3View:
    """lbm.u.x[n] / lbm.u.y[n] / lbm.u.z[n] (FX/lbm.hpp:359-372)"""

    def __init__(self, arr, N):
        self.x, self.y, self.z = arr[0:N], arr[N:2 * N], arr[2 * N:3 * N]


class _Field:
    def __init__(self, lbm, field, mask, arr, dims):
        self._lbm, self._field, self._mask, self.data = lbm, field, mask, arr
        if dims == 3:
            v = _Vec3View(arr, lbm.get_N())
            self.x, self.y, self.z = v.x, v.y, v.z

    def __getitem__(self, n): return self.data[n]
    def __setitem__(self, n, v): self.data[n] = v
    def __len__(self): return len(self.data)

    def read_from_device(self):  # Memory_Container::read_from_device, FX/lbm.hpp:406-412
        capi.check(self._lbm._L.luw_download(self._lbm._h, self._mask))

    def write_to_device(self):  # FX/lbm.hpp:413-416
        capi.check(self._lbm._L.luw_upload(self._lbm._h, self._mask))


class LBM:
    """LBM(Nx, Ny, Nz, nu, fx, fy, fz) -- FX/lbm.hpp:444-451 (sigma/alpha/beta of the SURFACE / TEMPERATURE
    extensions are not part of this path).  Keyword-only extras carry what the reference keeps in process
    globals or compile-time defines: DDF format, domain placement, nudging / sponge constants."""

    def __init__(self, Nx, Ny, Nz, nu, fx=0.0, fy=0.0, fz=0.0, *, fp16c=False, D=(1, 1, 1), O=(0, 0, 0),
                 force_field=False, update_fields_every_step=False, subgrid=True, device=0, kernel=capi.KERNEL_AUTO,
                 buffer_nudging=None, top_sponge=None, alpha=None, native_arith=False):
        self._L = capi.load()
        cfg = capi.Config()
        cfg.struct_size = C.sizeof(capi.Config)
        cfg.Nx, cfg.Ny, cfg.Nz = int(Nx), int(Ny), int(Nz)
        cfg.Dx, cfg.Dy, cfg.Dz = D
        cfg.Ox, cfg.Oy, cfg.Oz = O
        cfg.nu = float(nu)
        cfg.fx, cfg.fy, cfg.fz = float(fx), float(fy), float(fz)
        cfg.ddf_format = capi.DDF_FP16C if fp16c else capi.DDF_FP32
        cfg.options = (capi.OPT_FORCE_FIELD if force_field else 0) | (capi.OPT_UPDATE_FIELDS_EVERY_STEP if update_fields_every_step else 0) | (0 if subgrid
            else capi.OPT_NO_SUBGRID) | (capi.OPT_TEMPERATURE if alpha is not None else 0) | (capi.OPT_NATIVE_ARITH if native_arith else 0)
        cfg.alpha = float(alpha) if alpha is not None else 0.0      # thermal D3Q7 lattice: LBM(..., alpha, beta), FX/lbm.hpp:444
        if buffer_nudging is not None:  # dict(n_cells, inv_tau, downstream_face, nudge_vertical): FX/setup.cpp:3844-3866
            cfg.buffer_nudging_active = 1
            cfg.buffer_n_cells = int(buffer_nudging["n_cells"])
            cfg.buffer_inv_tau_lbmu = float(buffer_nudging["inv_tau"])
            cfg.buffer_downstream_face_id = int(buffer_nudging.get("downstream_face", 0))
            cfg.buffer_nudge_vertical = int(buffer_nudging.get("nudge_vertical", 0))
        if top_sponge is not None:  # dict(n_cells, inv_tau): FX/setup.cpp:3867-3903
            cfg.top_sponge_active = 1
            cfg.sponge_n_cells = int(top_sponge["n_cells"])
            cfg.sponge_inv_tau_lbmu = float(top_sponge["inv_tau"])
        cfg.device = int(device)
        cfg.kernel = int(kernel)
        self.cfg = cfg
        h = C.c_void_p()
        capi.check(self._L.luw_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.Nx, self.Ny, self.Nz = int(Nx), int(Ny), int(Nz)
        N = self.get_N()

        def view(field, ctype, count):
            ptr = self._L.luw_host_ptr(self._h, field)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)) if ptr else None
        self.rho = _Field(self, capi.FIELD_RHO, capi.MASK_RHO, view(capi.FIELD_RHO, C.c_float, N), 1)
        self.u = _Field(self, capi.FIELD_U, capi.MASK_U, view(capi.FIELD_U, C.c_float, 3 * N), 3)
        self.flags = _Field(self, capi.FIELD_FLAGS, capi.MASK_FLAGS, view(capi.FIELD_FLAGS, C.c_uint8, N), 1)
        Fv = view(capi.FIELD_F, C.c_float, 3 * N)
        self.F = _Field(self, capi.FIELD_F, capi.MASK_F, Fv, 3) if Fv is not None else None
        Tv = view(capi.FIELD_T, C.c_float, N)
        self.T = _Field(self, capi.FIELD_T, capi.MASK_T, Tv, 1) if Tv is not None else None
        self._initialized = False

    # ---- life cycle
    def close(self):
        if getattr(self, "_h", None):
            self._L.luw_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API
    def get_Nx(self): return self.Nx
    def get_Ny(self): return self.Ny
    def get_Nz(self): return self.Nz
    def get_N(self): return self.Nx * self.Ny * self.Nz
    def get_t(self): return int(self._L.luw_get_t(self._h))
    def index(self, x, y, z): return x + (y + z * self.Ny) * self.Nx  # FX/lbm.hpp:512-514
    def position(self, x, y, z):  # FX/lbm.hpp:523-525
        f = np.float32
        return (f(x) - f(0.5) * f(self.Nx) + f(0.5), f(y) - f(0.5) * f(self.Ny) + f(0.5), f(z) - f(0.5) * f(self.Nz) + f(0.5))

    def set_f(self, fx, fy, fz): capi.check(self._L.luw_set_f(self._h, fx, fy, fz))
    def set_coriolis(self, ox, oy, oz): capi.check(self._L.luw_set_coriolis(self._h, ox, oy, oz))

    def run(self, steps=0):
        """LBM::run (FX/lbm.cpp:1292-1312): first call uploads the host fields and runs `initialize`."""
        if not self._initialized:
            capi.check(self._L.luw_initialize(self._h))
            self._initialized = True
        if steps:
            capi.check(self._L.luw_run(self._h, int(steps)))

    def run_timed(self, steps):
        """like run(); returns the mean stream_collide kernel time in ms (HIP events on the launch stream)"""
        if not self._initialized:
            self.run(0)
        ms = C.c_double()
        capi.check(self._L.luw_run_timed(self._h, int(steps), C.byref(ms)))
        return ms.value

    def voxelize_mesh_on_device(self, tri, flag=capi.TYPE_S, bounds=None):
        """LBM::voxelize_mesh_on_device (FX/lbm.hpp:560): tri = float32 array (T,3,3) of triangle corners in lattice index
        coordinates; marks cells inside the mesh in self.flags (host mirror updated)"""
        tri = np.ascontiguousarray(tri, np.float32)
        p0, p1, p2 = (np.ascontiguousarray(tri[:, k, :]) for k in range(3))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        b = None if bounds is None else np.ascontiguousarray(bounds, np.float32).reshape(6)
        capi.check(self._L.luw_voxelize_mesh(self._h, tri.shape[0], p(p0), p(p1), p(p2), None if b is None else p(b), int(flag)))

    # ---- von-Karman inlet (tables built like FX/setup.cpp:886-1057; see host/vk_inlet.hpp)
    def vk_inlet_attach(self, point_cell, point_face, point_data, mode_data, mode_count, update_stride=1, stride_interpolation=False):
        pc = np.ascontiguousarray(point_cell, np.uint64); pf = np.ascontiguousarray(point_face, np.uint8)
        pd = np.ascontiguousarray(point_data, np.float32); md = np.ascontiguousarray(mode_data, np.float32)
        assert pd.size == 7 * pc.size and md.size == 50 * int(mode_count)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        capi.check(
            self._L.luw_vk_inlet_attach(self._h, pc.size, int(mode_count), p(pc), p(pf), p(pd), p(md), int(update_stride), int(bool(stride_interpolation))))

    def vk_inlet_apply(self): capi.check(self._L.luw_vk_inlet_apply(self._h))
    def vk_inlet_detach(self): capi.check(self._L.luw_vk_inlet_detach(self._h))

    # ---- on-device time averaging (replaces FX/setup.cpp:4441-4542)
    def stats_reset(self): capi.check(self._L.luw_stats_reset(self._h))
    def stats_accumulate(self): capi.check(self._L.luw_stats_accumulate(self._h))
    def run_sampled(self, steps, first_sample=1, stride=1):
        """`steps` steps of which number first_sample (from 1) and every stride-th after it are statistics samples
        (the purge_avg window of run_lbm, FX/setup.cpp:4252-4268)"""
        capi.check(self._L.luw_run_sampled(self._h, int(steps), int(first_sample), int(stride)))
    def stats_download(self):
        N = self.get_N()
        out = dict(avg_u=np.zeros(3 * N, np.float32), avg_rho=np.zeros(N, np.float32), m2_u=np.zeros(N, np.float32),
                   m2_v=np.zeros(N, np.float32), m2_w=np.zeros(N, np.float32))
        cnt = C.c_uint64(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        capi.check(self._L.luw_stats_download(self._h, p(out["avg_u"]), p(out["avg_rho"]), p(out["m2_u"]), p(out["m2_v"]), p(out["m2_w"]), C.byref(cnt)))
        out["count"] = cnt.value
        return out

    def stats_download_T(self):
        out = np.zeros(self.get_N(), np.float32)
        capi.check(self._L.luw_stats_download_T(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- probe columns: u at a short list of cells per step (luw_gather_*)
    def gather_attach(self, cells):
        c = np.ascontiguousarray(cells, np.uint64)
        self._gather_n = int(c.size)
        capi.check(self._L.luw_gather_attach(self._h, c.size, c.ctypes.data_as(C.c_void_p)))

    def gather_u(self):
        out = np.zeros((getattr(self, "_gather_n", 0), 3), np.float32)
        capi.check(self._L.luw_gather_u(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- device-level interface (multi-GPU driver)
    def area(self, direction): return int(self._L.luw_get_area(self._h, direction))
    def set_stream(self, stream_ptr): capi.check(self._L.luw_set_stream(self._h, stream_ptr))
    def enqueue_stream_collide(self, box, write_fields=False, sample=False):
        x0, x1, y0, y1, z0, z1 = box
        capi.check(self._L.luw_enqueue_stream_collide(self._h, x0, x1, y0, y1, z0, z1, int(bool(write_fields)) | (2 if sample else 0)))

    def stats_begin_sample(self):
        """counts a sampled step; True when its boxes can carry the statistics themselves (enqueue_stream_collide(..., sample=True))"""
        fused = C.c_int32(0)
        capi.check(self._L.luw_stats_begin_sample(self._h, C.byref(fused)))
        return bool(fused.value)
    def enqueue_extract_fi(self, direction, buf_p_ptr, buf_m_ptr): capi.check(self._L.luw_enqueue_extract_fi(self._h, direction, buf_p_ptr, buf_m_ptr))
    def set_x_face_buffers(self, buf_p_ptr, buf_m_ptr): capi.check(self._L.luw_set_x_face_buffers(self._h, buf_p_ptr, buf_m_ptr))
    def set_x_face_inputs(self, buf_p_ptr, buf_m_ptr): capi.check(self._L.luw_set_x_face_inputs(self._h, buf_p_ptr, buf_m_ptr))
    def edge_length(self, e): return int(self._L.luw_get_edge_length(self._h, e))

    def enqueue_edges(self, ptrs, insert):
        """the 12 edge buffers (device pointers, 0 where the domain has no such edge): pack (insert False) or unpack"""
        arr = (C.c_void_p * 12)(*[C.c_void_p(p) if p else None for p in ptrs])
        capi.check((self._L.luw_enqueue_insert_edges if insert else self._L.luw_enqueue_extract_edges)(self._h, arr))
    def enqueue_insert_fi(self, direction, buf_p_ptr, buf_m_ptr): capi.check(self._L.luw_enqueue_insert_fi(self._h, direction, buf_p_ptr, buf_m_ptr))
    def enqueue_extract_gi(self, direction, buf_p_ptr, buf_m_ptr): capi.check(self._L.luw_enqueue_extract_gi(self._h, direction, buf_p_ptr, buf_m_ptr))
    def enqueue_insert_gi(self, direction, buf_p_ptr, buf_m_ptr): capi.check(self._L.luw_enqueue_insert_gi(self._h, direction, buf_p_ptr, buf_m_ptr))
    def increment_time_step(self, steps=1): capi.check(self._L.luw_increment_time_step(self._h, steps))
    def reset_time_step(self): capi.check(self._L.luw_reset_time_step(self._h))
    # ---- the library's schedule of one domain's share of a decomposed step (luw_domain_step_*, csrc/luw_group.hpp)
    def domain_step_create(self, compute_stream, comm_stream, x_shell=0, overlap=True):
        d = C.c_void_p()
        capi.check(self._L.luw_domain_step_create(self._h, compute_stream, comm_stream, int(x_shell), int(bool(overlap)), C.byref(d)))
        return d
    def domain_step_destroy(self, d): self._L.luw_domain_step_destroy(d)
    def domain_step_overlaps(self, d): return bool(self._L.luw_domain_step_overlaps(d))
    def domain_step_launch(self, d, write_fields, timed=False): capi.check(self._L.luw_domain_step_launch(d, int(write_fields), int(bool(timed))))
    def domain_step_separate_stats(self, d): capi.check(self._L.luw_domain_step_separate_stats(d))
    def domain_step_timing(self, d):
        k, sh = C.c_double(0.0), C.c_double(0.0)
        capi.check(self._L.luw_domain_step_timing(d, C.byref(k), C.byref(sh)))
        return k.value, sh.value
    def fields_every_step(self): return bool(self._L.luw_fields_every_step(self._h))   # option, or a fluid nudging / sponge reference cell (luw_initialize)
    def finish(self): capi.check(self._L.luw_finish(self._h))
    def device_ptr(self, field): return self._L.luw_device_ptr(self._h, field)
    def pitch(self): return int(self._L.luw_get_pitch(self._h))
    def plane_stride(self): return int(self._L.luw_get_plane_stride(self._h))
    def download_gi(self):
        """thermal DDFs as stored, reference layout gi[i*N+n], i = 0..6"""
        out = np.zeros(7 * self.get_N(), np.uint16 if self.cfg.ddf_format == capi.DDF_FP16C else np.float32)
        capi.check(self._L.luw_download_gi(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def download_fi(self):
        """DDFs as stored (float32 values or uint16 FP16C codes), reference layout fi[i*N+n]"""
        out = np.zeros(19 * self.get_N(), np.uint16 if self.cfg.ddf_format == capi.DDF_FP16C else np.float32)
        capi.check(self._L.luw_download_fi(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def upload_fi(self, fi):
        fi = np.ascontiguousarray(fi, np.uint16 if self.cfg.ddf_format == capi.DDF_FP16C else np.float32)
        assert fi.size == 19 * self.get_N()
        capi.check(self._L.luw_upload_fi(self._h, fi.ctypes.data_as(C.c_void_p)))

    def ddf_itemsize(self): return 2 if self.cfg.ddf_format == capi.DDF_FP16C else 4


class LBMGroup:
    """The reference's multi-domain `LBM(N, Dx, Dy, Dz, nu, ...)` (FX/lbm.hpp:444-450) in ONE process over the luw_group_* C-ABI: all
    D domains, their streams and the halo traffic between them live in the library; this mirror holds GLOBAL host arrays in the
    reference layout (flags[n], u.x[n] ... with n over the whole lattice, like Memory_Container's global index space) which are
    scattered to / gathered from the domains' mirrors.  devices: HIP device per domain (None: one device each, cfg.device + d)."""

    def __init__(self, Nx, Ny, Nz, Dx, Dy, Dz, nu, fx=0.0, fy=0.0, fz=0.0, *, fp16c=False, devices=None, force_field=False, update_fields_every_step=False,
                 subgrid=True, device=0, kernel=capi.KERNEL_AUTO, buffer_nudging=None, top_sponge=None, alpha=None, global_arrays=True, native_arith=False):
        self._L = capi.load()
        cfg = capi.Config()
        cfg.struct_size = C.sizeof(capi.Config)
        cfg.Nx, cfg.Ny, cfg.Nz = int(Nx), int(Ny), int(Nz)
        cfg.Dx, cfg.Dy, cfg.Dz = int(Dx), int(Dy), int(Dz)
        cfg.nu = float(nu)
        cfg.fx, cfg.fy, cfg.fz = float(fx), float(fy), float(fz)
        cfg.ddf_format = capi.DDF_FP16C if fp16c else capi.DDF_FP32
        cfg.options = (capi.OPT_FORCE_FIELD if force_field else 0) | (capi.OPT_UPDATE_FIELDS_EVERY_STEP if update_fields_every_step else 0) | (0 if subgrid
            else capi.OPT_NO_SUBGRID) | (capi.OPT_TEMPERATURE if alpha is not None else 0) | (capi.OPT_NATIVE_ARITH if native_arith else 0)
        cfg.alpha = float(alpha) if alpha is not None else 0.0
        if buffer_nudging is not None:
            cfg.buffer_nudging_active = 1
            cfg.buffer_n_cells = int(buffer_nudging["n_cells"]); cfg.buffer_inv_tau_lbmu = float(buffer_nudging["inv_tau"])
            cfg.buffer_downstream_face_id = int(buffer_nudging.get("downstream_face", 0)); cfg.buffer_nudge_vertical = int(
                buffer_nudging.get("nudge_vertical", 0))
        if top_sponge is not None:
            cfg.top_sponge_active = 1
            cfg.sponge_n_cells = int(top_sponge["n_cells"]); cfg.sponge_inv_tau_lbmu = float(top_sponge["inv_tau"])
        cfg.device = int(device); cfg.kernel = int(kernel)
        self.cfg = cfg
        n = int(Dx) * int(Dy) * int(Dz)
        dev = None
        if devices is not None:
            assert len(devices) == n
            dev = (C.c_int * n)(*[int(d) for d in devices])
        h = C.c_void_p()
        capi.check(self._L.luw_group_create(C.byref(cfg), dev, C.byref(h)))
        self._h = h
        self.Nx, self.Ny, self.Nz, self.D = int(Nx), int(Ny), int(Nz), (int(Dx), int(Dy), int(Dz))
        N = self.Nx * self.Ny * self.Nz
        if global_arrays:
            self.flags = np.zeros(N, np.uint8); self.u = np.zeros(3 * N, np.float32); self.rho = np.ones(N, np.float32)
            self.T = np.ones(N, np.float32) if alpha is not None else None
            self.F = np.zeros(3 * N, np.float32) if force_field else None
        else:               # the caller works on the domains' own mirrors (domain_host / initialize_from_domains)
            self.flags = self.u = self.rho = self.T = self.F = None
        self._initialized = False

    def close(self):
        if getattr(self, "_h", None):
            self._L.luw_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _p(self, a): return a.ctypes.data_as(C.c_void_p)
    def get_N(self): return self.Nx * self.Ny * self.Nz
    def get_t(self): return int(self._L.luw_group_get_t(self._h))
    def get_D(self): return self.D[0] * self.D[1] * self.D[2]
    def overlaps(self): return bool(self._L.luw_group_overlaps(self._h))
    def direct_peer_stores(self): return bool(self._L.luw_group_direct_peer_stores(self._h))
    def one_phase(self): return bool(self._L.luw_group_one_phase(self._h))
    def transport(self): return int(self._L.luw_group_transport(self._h))   # capi.TRANSPORT_NAMES
    def set_f(self, fx, fy, fz): capi.check(self._L.luw_group_set_f(self._h, fx, fy, fz))
    def set_coriolis(self, ox, oy, oz): capi.check(self._L.luw_group_set_coriolis(self._h, ox, oy, oz))

    def domain_info(self, d):
        lN = (C.c_uint32 * 3)(); off = (C.c_int32 * 3)(); dev = C.c_int(0)
        capi.check(self._L.luw_group_domain_info(self._h, d, lN, off, C.byref(dev)))
        return tuple(lN), tuple(off), dev.value

    def domain_host(self, d):
        """(flags, u, rho) host mirrors of domain d -- local box incl. halo layers, reference layout -- for callers that fill the domains
        directly (lbm.lbm_domain[d]->flags ... of the reference) instead of going through the global arrays"""
        lN, _, _ = self.domain_info(d)
        n = lN[0] * lN[1] * lN[2]
        s = self._L.luw_group_domain(self._h, d)
        view = lambda field, ct, count: np.ctypeslib.as_array(C.cast(self._L.luw_host_ptr(s, field), C.POINTER(ct)), shape=(count,))
        return view(capi.FIELD_FLAGS, C.c_uint8, n), view(capi.FIELD_U, C.c_float, 3 * n), view(capi.FIELD_RHO, C.c_float, n)

    def initialize_from_domains(self):
        """upload + initialise from the domains' own host mirrors (filled through domain_host): no global arrays, no scatter"""
        capi.check(self._L.luw_group_initialize(self._h))
        self._initialized = True

    def write_to_device(self):
        """the global host arrays -> every domain's mirror (halo layers included) -> device"""
        for field, arr in ((capi.FIELD_FLAGS, self.flags), (capi.FIELD_U, self.u), (capi.FIELD_RHO, self.rho), (capi.FIELD_T, self.T), (capi.FIELD_F, self.F)):
            if arr is not None:
                capi.check(self._L.luw_group_scatter(self._h, field, self._p(arr)))

    def read_from_device(self, fields=("u", "rho")):
        mask = {"u": capi.MASK_U, "rho": capi.MASK_RHO, "flags": capi.MASK_FLAGS, "T": capi.MASK_T}
        ids = {"u": capi.FIELD_U, "rho": capi.FIELD_RHO, "flags": capi.FIELD_FLAGS, "T": capi.FIELD_T}
        m = 0
        for f in fields: m |= mask[f]
        capi.check(self._L.luw_group_download(self._h, m))
        for f in fields:
            capi.check(self._L.luw_group_gather(self._h, ids[f], self._p(getattr(self, f))))

    def run(self, steps=0):
        if not self._initialized:
            self.write_to_device()
            capi.check(self._L.luw_group_initialize(self._h))
            self._initialized = True
        if steps:
            capi.check(self._L.luw_group_run(self._h, int(steps)))

    def run_sampled(self, steps, first_sample=1, stride=1):
        capi.check(self._L.luw_group_run_sampled(self._h, int(steps), int(first_sample), int(stride)))

    def run_timed(self, steps):
        if not self._initialized:
            self.run(0)
        ms = C.c_double()
        capi.check(self._L.luw_group_run_timed(self._h, int(steps), C.byref(ms)))
        return ms.value

    def voxelize_mesh_on_device(self, tri, flag=capi.TYPE_S, bounds=None):
        """every domain voxelises its own box; the result is gathered into self.flags"""
        tri = np.ascontiguousarray(tri, np.float32)
        p0, p1, p2 = (np.ascontiguousarray(tri[:, k, :]) for k in range(3))
        b = None if bounds is None else np.ascontiguousarray(bounds, np.float32).reshape(6)
        capi.check(self._L.luw_group_scatter(self._h, capi.FIELD_FLAGS, self._p(self.flags)))
        capi.check(self._L.luw_group_scatter(self._h, capi.FIELD_U, self._p(self.u)))
        capi.check(self._L.luw_group_voxelize_mesh(self._h, tri.shape[0], self._p(p0), self._p(p1), self._p(p2), None if b is None else self._p(b), int(flag)))
        capi.check(self._L.luw_group_gather(self._h, capi.FIELD_FLAGS, self._p(self.flags)))

    def vk_inlet_attach(self, point_cell, point_face, point_data, mode_data, mode_count, update_stride=1, stride_interpolation=False):
        pc = np.ascontiguousarray(point_cell, np.uint64); pf = np.ascontiguousarray(point_face, np.uint8)
        pd = np.ascontiguousarray(point_data, np.float32); md = np.ascontiguousarray(mode_data, np.float32)
        capi.check(self._L.luw_group_vk_inlet_attach(self._h, pc.size, int(mode_count), self._p(pc), self._p(pf), self._p(pd), self._p(md), int(update_stride),
            int(bool(stride_interpolation))))

    def gather_attach(self, cells):
        c = np.ascontiguousarray(cells, np.uint64)
        self._gather_n = int(c.size)
        capi.check(self._L.luw_group_gather_attach(self._h, c.size, self._p(c)))

    def gather_u(self):
        out = np.zeros((getattr(self, "_gather_n", 0), 3), np.float32)
        capi.check(self._L.luw_group_gather_u(self._h, self._p(out)))
        return out

    def stats_reset(self): capi.check(self._L.luw_group_stats_reset(self._h))

    def stats_download(self):
        N = self.get_N()
        out = dict(avg_u=np.zeros(3 * N, np.float32), avg_rho=np.zeros(N, np.float32), m2_u=np.zeros(N, np.float32), m2_v=np.zeros(N, np.float32),
            m2_w=np.zeros(N, np.float32))
        if self.T is not None:
            out["avg_T"] = np.zeros(N, np.float32)
        cnt = C.c_uint64(0)
        capi.check(self._L.luw_group_stats_download(self._h, self._p(out["avg_u"]), self._p(out["avg_rho"]), self._p(out["m2_u"]), self._p(out["m2_v"]),
            self._p(out["m2_w"]),
                                                    self._p(out["avg_T"]) if self.T is not None else None, C.byref(cnt)))
        out["count"] = cnt.value
        return out

    def download_fi_domain(self, d):
        """DDFs of domain d as stored (local reference layout incl. halos): test access"""
        lN, _, _ = self.domain_info(d)
        out = np.zeros(19 * lN[0] * lN[1] * lN[2], np.uint16 if self.cfg.ddf_format == capi.DDF_FP16C else np.float32)
        capi.check(self._L.luw_download_fi(self._L.luw_group_domain(self._h, d), self._p(out)))
        return out
