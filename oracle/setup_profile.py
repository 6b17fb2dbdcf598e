"""CPU restatement of the reference's HOST-side set-up for profile mode (*.luwpf).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/core/cfd_core/FluidX3D/src ("FX/") in FP32, op by op:
  deck grammar              FX/setup.cpp:40-178, key handlers :2918-3305, mesh_control :3364-3390
  grid sizing / sponge ext  FX/setup.cpp:3552-3568
  profile.dat reader        FX/setup.cpp:2122-2150, sample clean-up :3681-3723
  Units                     FX/units.hpp:21-28,44-67; call site FX/setup.cpp:3731-3737
  buffer / sponge constants FX/setup.cpp:3844-3903
  STL load + transform      FX/utilities.hpp:4835-4888,4774-4804; FX/setup.cpp:4070-4087
  profile table             FX/setup.cpp:5777-5782,5847-5879, Hermite interpolation :2243-2280,
                            FX/utilities.hpp:2374-2377
  nearest-index lookup      FX/setup.cpp:5901-5912
  voxelisation along z      FX/kernel.cpp:2381-2471 (single pass, direction 2: FX/lbm.cpp:1427-1430), bbox
                            pad FX/lbm.cpp:498
  flags / u fill            FX/setup.cpp:5914-5995, wind direction :6009-6013, downstream face :3749-3761
  VTK scaling               FX/lbm.hpp:307-356

Used by tests to (a) reproduce the initial state of the committed real-reference runs, (b) check the
product's C++ driver against the same numbers.
"""
import ctypes
import ctypes.util
import math
import os
import struct

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m"))
_libm.sinf.argtypes = [ctypes.c_float]; _libm.sinf.restype = ctypes.c_float
_libm.cosf.argtypes = [ctypes.c_float]; _libm.cosf.restype = ctypes.c_float
_libm.lroundf.argtypes = [ctypes.c_float]; _libm.lroundf.restype = ctypes.c_long

TYPE_S, TYPE_E = 0x01, 0x02


def lround(x):
    return int(_libm.lroundf(float(f32(x))))


# ----------------------------------------------------------------------------- deck
def _comment_index(line):
    in_s = in_d = False
    for i in range(len(line) - 1):
        ch, nx = line[i], line[i + 1]
        if ch == "'" and not in_d:
            in_s = not in_s; continue
        if ch == '"' and not in_s:
            in_d = not in_d; continue
        if not in_s and not in_d and ch == "/" and nx == "/":
            return i
    return -1


def _normalize_key(key):
    key = key.strip(" \t\r\n")
    out, last_sep = [], False
    for ch in key:
        if ch == "-" or ch.isspace():
            if out and not last_sep:
                out.append("_")
            last_sep = True
            continue
        out.append(ch.lower()); last_sep = False
    s = "".join(out).strip("_")
    return {"vk_inlet_enable": "turb_inflow_enable", "vk_inlet_anisotropy_scale": "vk_inlet_anisotropy",
            "vk_inlet_aniso_scale": "vk_inlet_anisotropy"}.get(s, s)


def unquote(s):
    s = s.strip(" \t\r\n")
    if len(s) >= 2 and s[0] in "\"'" and s[-1] == s[0]:
        s = s[1:-1].strip(" \t\r\n")
    return s


def read_deck(path):
    vals = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            c = _comment_index(line)
            if c >= 0:
                line = line[:c]
            eq = line.find("=")
            if eq < 0:
                continue
            k = _normalize_key(line[:eq])
            if k:
                vals[k] = line[eq + 1:].strip(" \t\r\n")
    return vals


_BOOL = {"1": True, "true": True, "t": True, "yes": True, "y": True, "on": True, "enable": True, "enabled": True,
         "0": False, "false": False, "f": False, "no": False, "n": False, "off": False, "disable": False,
         "disabled": False}


def parse_bool(raw, default):
    s = unquote(raw).lower()
    if not s:
        return default
    if s in _BOOL:
        return _BOOL[s]
    try:
        return float(s) != 0.0
    except ValueError:
        return default


def _atof(s):
    """C atof on the leading numeric prefix"""
    import re
    m = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", s)
    return float(m.group(0)) if m else 0.0


def _second_val(rng):
    c = rng.find(","); r = rng.find("]", c)
    return f32(_atof(rng[c + 1:r]))


def _float_list(rng):
    s = rng.strip()
    lb, rb = s.find("["), s.find("]")
    inside = s[lb + 1:rb] if (lb >= 0 and rb > lb) else s
    return [f32(_atof(t)) for t in inside.split(",") if t.strip()]


# ----------------------------------------------------------------------------- units
class Units:
    def set_m_kg_s_K(self, x, u, rho, T, si_x, si_u, si_rho, si_T):
        self.unit_m = f32(si_x) / f32(x)
        self.unit_kg = f32(si_rho) / f32(rho) * (self.unit_m * self.unit_m * self.unit_m)
        self.unit_s = f32(u) / f32(si_u) * self.unit_m
        self.unit_K = f32(si_T) / f32(T)

    def x(self, si_x): return f32(si_x) / self.unit_m
    def si_x(self, x): return f32(x) * self.unit_m
    def nu(self, si_nu): return f32(si_nu) * self.unit_s / (self.unit_m * self.unit_m)
    def si_u(self, u): return f32(u) * self.unit_m / self.unit_s
    def si_rho(self, rho): return f32(rho) * self.unit_kg / (self.unit_m * self.unit_m * self.unit_m)
    def t(self, si_t): return int(f32(si_t) / self.unit_s)


# ----------------------------------------------------------------------------- profile
def read_profile_dat(path):
    out = []
    with open(path) as f:
        for line in f:
            c = line.find("//")
            if c >= 0: line = line[:c]
            c = line.find("#")
            if c >= 0: line = line[:c]
            line = line.strip()
            if not line: continue
            tok = line.replace(",", " ").replace(";", " ").split()
            try:
                z, u = f32(float(tok[0])), f32(float(tok[1]))
            except (ValueError, IndexError):
                continue
            if np.isfinite(z) and np.isfinite(u):
                out.append((z, u))
    return out


def hermite_spline(a, b, va, vb, t):
    cbt, sqt = t * t * t, t * t
    return (f32(2.0) * cbt - f32(3.0) * sqt + f32(1.0)) * a + (f32(-2.0) * cbt + f32(3.0) * sqt) * b + (cbt - f32(2.0) * sqt + t) * va + (cbt - sqt) * vb


def interpolate_profile_cubic(z, u, zq):
    n = len(z)
    if n == 0: return f32(0)
    if n == 1: return u[0]
    if zq <= z[0]: return u[0]
    if zq >= z[-1]: return u[-1]
    it = int(np.searchsorted(np.array(z, f32), zq, side="right"))  # upper_bound
    i1 = 0 if it == 0 else it - 1
    i2 = min(i1 + 1, n - 1)
    z0, z1 = z[i1], z[i2]
    denom = z1 - z0
    if denom <= 0: return u[i1]
    t = (zq - z0) / denom

    def slope_at(i):
        if i == 0:
            dz = z[1] - z[0]; return (u[1] - u[0]) / dz if dz != 0 else f32(0)
        if i + 1 >= n:
            dz = z[n - 1] - z[n - 2]; return (u[n - 1] - u[n - 2]) / dz if dz != 0 else f32(0)
        dz = z[i + 1] - z[i - 1]; return (u[i + 1] - u[i - 1]) / dz if dz != 0 else f32(0)
    m0, m1 = slope_at(i1), slope_at(i2)
    return hermite_spline(u[i1], u[i2], m0 * denom, m1 * denom, t)


# ----------------------------------------------------------------------------- STL + voxeliser
def read_stl(path):
    data = open(path, "rb").read()
    nt = struct.unpack_from("<I", data, 80)[0]
    assert len(data) == 84 + 50 * nt, "only binary STL is supported (FX/utilities.hpp:4846-4849)"
    tri = np.zeros((nt, 3, 3), f32)
    for i in range(nt):
        v = struct.unpack_from("<12f", data, 84 + 50 * i)
        tri[i] = np.array(v[3:12], f32).reshape(3, 3)
    return tri


def voxelize_z(tri, Nx, Ny, Nz, flags):
    """FX/kernel.cpp:2381-2471 with direction=2, flag=TYPE_S, single domain (O=0, offset maps to global index
    space: position(xyz)+offset == xyz)."""
    pmin = tri.reshape(-1, 3).min(axis=0); pmax = tri.reshape(-1, 3).max(axis=0)
    x0, y0, z0 = (pmin - f32(2.0)); x1, y1, z1 = (pmax + f32(2.0))

    def clampi(v, lo, hi): return max(lo, min(hi, v))
    zstart = clampi(int(z0), 0, Nz - 1)
    hmax = clampi(int(z1), 0, Nz)
    p0, p1, p2 = tri[:, 0, :], tri[:, 1, :], tri[:, 2, :]
    uu = p1 - p0; vv = p2 - p0
    rd = np.array([0, 0, 1], f32)
    hh = np.cross(np.broadcast_to(rd, vv.shape), vv).astype(f32)   # cross(r_direction, v)
    g = np.einsum("ij,ij->i", uu, hh).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        fi = (f32(1.0) / g).astype(f32)
    for y in range(Ny):
        for x in range(Nx):
            ro = np.array([x, y, zstart], f32)
            if ro[0] < x0 or ro[1] < y0 or ro[0] >= x1 or ro[1] >= y1:
                continue
            w = ro - p0
            q = np.cross(w, uu).astype(f32)
            with np.errstate(invalid="ignore"):
                s = fi * np.einsum("ij,ij->i", w, hh).astype(f32)
                t = fi * (q @ rd).astype(f32)
                d = fi * np.einsum("ij,ij->i", vv, q).astype(f32)
                hit = (g != 0) & (s >= 0) & (s < 1) & (t >= 0) & (s + t < 1)
            ahead = hit & (d > 0)
            intersections = int(ahead.sum())
            intersections_check = int((hit & ~(d > 0)).sum())
            dist = sorted(int(v) for v in d[ahead][:64] if v < 65536.0)
            # the kernel stores (ushort)d for the first 64 hits in triangle order, then sorts
            dist = sorted([int(v) & 0xFFFF for v in d[ahead]][:64])
            inside = bool(intersections % 2) and bool(intersections_check % 2)
            intersection = int(intersections % 2 != intersections_check % 2)
            h0 = zstart
            hmesh = h0 + (dist[min(intersections - 1, 63)] if intersections > 0 and dist else 0)
            for h in range(h0, hmax):
                while intersection < intersections and h > h0 + dist[min(intersection, 63)]:
                    inside = not inside
                    intersection += 1
                inside = inside and (intersection < intersections and h < hmesh)
                n = x + (y + h * Ny) * Nx
                if inside:
                    flags[n] = (int(flags[n]) & 0xFC) | TYPE_S
                # "outside" branch only clears cells that were solid before; nothing is solid yet here


# ----------------------------------------------------------------------------- main entry
def setup_profile_case(deck_path, angle_index=0, solid_mask=None):
    """Returns a dict with everything LBM::LBM + the profile-mode fill produce for one angle.

    solid_mask (optional, bool[Nz_core,Ny,Nx]): use this TYPE_S mask instead of running the voxeliser.  Faces
    of LUW geometry sit on exact lattice planes by construction (pmin -> 1.0, FX/setup.cpp:4087), where the
    GPU voxeliser's result depends on the device reciprocal rounding (FX/kernel.cpp:2406,2409); the committed
    real-reference fixtures therefore carry the mask the reference itself produced."""
    parent = os.path.dirname(os.path.abspath(deck_path))
    dk = read_deck(deck_path)
    g = lambda k, d="": unquote(dk.get(k, d))
    case = g("casename", "example")
    si_size = [_second_val(dk[k]) if g(k) else f32(0) for k in ("si_x_cfd", "si_y_cfd", "si_z_cfd")]
    z_si_offset = f32(_atof(dk["base_height"])) if g("base_height") else f32(50.0)
    D = (1, 1, 1)
    if g("n_gpu"):
        ins = dk["n_gpu"][dk["n_gpu"].find("[") + 1:dk["n_gpu"].find("]")].split(",")
        if len(ins) == 3: D = tuple(max(1, int(_atof(t))) for t in ins)
    enable_buffer_nudging = parse_bool(dk.get("enable_buffer_nudging", ""), True)
    buffer_thickness_m = f32(_atof(g("buffer_thickness_m"))) if g("buffer_thickness_m") else f32(160.0)
    buffer_tau_s = f32(_atof(g("buffer_tau_s"))) if g("buffer_tau_s") else f32(300.0)
    buffer_nudge_vertical = 1 if parse_bool(dk.get("buffer_nudge_vertical", ""), False) else 0
    enable_top_sponge = parse_bool(dk.get("enable_top_sponge", ""), True)
    sponge_thickness_m = f32(_atof(g("sponge_thickness_m"))) if g("sponge_thickness_m") else f32(200.0)
    sponge_tau_s = f32(_atof(g("sponge_tau_s"))) if g("sponge_tau_s") else f32(120.0)
    run_nstep = int(_atof(g("run_nstep"))) if g("run_nstep") else 0
    mesh_control = g("mesh_control")
    cell_m = f32(20.0)
    if mesh_control == "cell_size" and g("cell_size"):
        cs = f32(_atof(g("cell_size")))
        if cs > 0 and np.isfinite(cs): cell_m = cs
    elif mesh_control == "gpu_memory":
        raise NotImplementedError("gpu_memory bisection is restated in the C++ driver tests only")
    angles = _float_list(dk["angle"])

    lbm_ref_u = f32(0.10); si_nu = f32(1.48E-5); si_rho = f32(1.225)
    Nx = max(1, int(si_size[0] / cell_m + f32(0.5)))
    Ny = max(1, int(si_size[1] / cell_m + f32(0.5)))
    sponge_cells_cfg = max(1, lround(sponge_thickness_m / cell_m))
    Nz_core = max(1, int(si_size[2] / cell_m + f32(0.5)))
    top_sponge_grid_extend = enable_top_sponge and sponge_tau_s > 0 and Nz_core > 2
    Nz = Nz_core + (sponge_cells_cfg if top_sponge_grid_extend else 0)
    side_ref_z_cap = Nz_core - 1 if top_sponge_grid_extend else -1

    # profile samples
    smp = sorted(read_profile_dat(os.path.join(parent, "wind_bc", "profile.dat")), key=lambda s: s[0])
    zv, uv = [], []
    for z, u in smp:
        if zv and abs(float(z - zv[-1])) < 1e-6:
            uv[-1] = u; continue
        zv.append(z); uv.append(u)
    domain_agl = si_size[2] - z_si_offset
    if domain_agl > 1.0 and zv[-1] <= 1.5:
        zv = [z * domain_agl for z in zv]
    si_ref_u = max(uv)

    units = Units()
    units.set_m_kg_s_K(f32(Ny), lbm_ref_u, f32(1), f32(1), si_size[1], si_ref_u, si_rho, f32(293.15))
    u_scale = lbm_ref_u / si_ref_u
    lbm_nu = units.nu(si_nu)

    # buffer nudging / sponge (FX/setup.cpp:3844-3903)
    min_dim = min(Nx, Ny, Nz); max_nbuf = max(1, min_dim // 4)
    nbuf = min(max(lround(buffer_thickness_m / cell_m), 1), max_nbuf)
    dt_si = cell_m * (lbm_ref_u / si_ref_u)
    buffer_inv_tau = dt_si / buffer_tau_s if buffer_tau_s > 0 else f32(0)
    buffer_active = enable_buffer_nudging and buffer_tau_s > 0
    ns = max(sponge_cells_cfg, 1)
    if Nz > 2: ns = min(ns, Nz - 2)
    sponge_inv_tau = dt_si / sponge_tau_s if sponge_tau_s > 0 else f32(0)
    sponge_active = top_sponge_grid_extend and sponge_tau_s > 0 and Nz_core > 2

    # STL (FX/setup.cpp:4001-4087): first match of <case>_DEM_PF.stl, <case>_DG.stl, *_DEM_PF.stl, *_DG.stl, *.stl
    pt = os.path.join(parent, "proj_temp")
    cands = [os.path.join(pt, case + "_DEM_PF.stl"), os.path.join(pt, case + "_DG.stl")]
    stl = next((c for c in cands if os.path.isfile(c)), None)
    if stl is None:
        names = sorted(os.listdir(pt))
        for suf in ("_DEM_PF.stl", "_DG.stl", ".stl"):
            m = [n for n in names if n.endswith(suf)]
            if m: stl = os.path.join(pt, m[0]); break
    tri = read_stl(stl)
    pts = np.concatenate([tri[0, :1, :], tri[1:].reshape(-1, 3)])   # find_bounds() seeds with p0[0] only (FX/utilities.hpp:4774-4785)
    stl_min, stl_max = pts.min(axis=0), pts.max(axis=0)
    stl_size = stl_max - stl_min
    domain_min_si = np.array([units.si_x(f32(0.5) - f32(0.5) * f32(n)) for n in (Nx, Ny, Nz)], f32)
    vtk_origin_shift = stl_min - domain_min_si
    scale_geom = units.x(si_size[0]) / stl_size[0]
    tri = (scale_geom * tri).astype(f32)                      # center = 0
    pmin = (scale_geom * stl_min).astype(f32)
    tri = (tri + (f32(1.0) - pmin)).astype(f32)               # translate so that pmin = (1,1,1)

    # profile table (FX/setup.cpp:5777-5879)
    origin = np.array([f32(0.5) - f32(0.5) * f32(n) for n in (Nx, Ny, Nz)], f32)
    flat_ground = origin[2] + units.x(z_si_offset)
    profile_dz = f32(0.1)
    solver_top_si = units.si_x(f32(Nz - 1))
    core_top_si = units.si_x(f32(side_ref_z_cap)) if side_ref_z_cap >= 0 else solver_top_si
    ground_min_si = units.si_x(flat_ground - origin[2])
    table_top = solver_top_si - ground_min_si
    if not np.isfinite(table_top) or table_top <= 0:
        table_top = max(profile_dz, si_size[2] - ground_min_si)
    table_top = max(table_top, profile_dz)
    steps = int(math.ceil(float(table_top / profile_dz)))
    prof_si = []
    for i in range(steps + 1):
        zq = min(table_top, f32(i) * profile_dz)
        v = interpolate_profile_cubic(zv, uv, zq)
        prof_si.append(v if v >= 0 else f32(0))
    prof_lbm = np.array([v * u_scale for v in prof_si], f32)
    inv_dz = f32(1.0) / profile_dz

    def speed(pos_z, ground_z):
        if pos_z <= ground_z: return f32(0)
        z_agl = units.si_x(pos_z - ground_z)
        if z_agl < 0: z_agl = f32(0)
        idx = max(0, lround(z_agl * inv_dz))
        return prof_lbm[min(idx, len(prof_lbm) - 1)]

    angle_deg = angles[angle_index]
    deg2rad = f32(3.14159265358979323846) / f32(180.0)
    angle_rad = angle_deg * deg2rad
    dir_x = f32(-_libm.sinf(float(angle_rad))); dir_y = f32(-_libm.cosf(float(angle_rad)))
    if abs(dir_x) >= abs(dir_y): dbc = "+x" if dir_x >= 0 else "-x"
    else: dbc = "+y" if dir_y >= 0 else "-y"
    buffer_face = {"-x": 1, "+x": 2, "-y": 3, "+y": 4}[dbc]

    N = Nx * Ny * Nz
    flags = np.zeros(N, np.uint8); u = np.zeros(3 * N, f32); rho = np.ones(N, f32)
    if solid_mask is None:
        voxelize_z(tri, Nx, Ny, Nz, flags)
    else:
        sm = np.zeros((Nz, Ny, Nx), bool); sm[:solid_mask.shape[0]] = solid_mask
        flags[sm.ravel()] = TYPE_S
    n_solid_vox = int((flags & TYPE_S).astype(bool).sum())

    posz = np.array([f32(z) - f32(0.5) * f32(Nz) + f32(0.5) for z in range(Nz)], f32)
    spd = np.array([speed(pz, flat_ground) for pz in posz], f32)
    fl3 = flags.reshape(Nz, Ny, Nx)
    u3 = u.reshape(3, Nz, Ny, Nx)
    # initialize_profile_velocity
    fluid = (fl3 & TYPE_S) == 0
    u3[0] = np.where(fluid, (dir_x * spd)[:, None, None], f32(0))
    u3[1] = np.where(fluid, (dir_y * spd)[:, None, None], f32(0))
    u3[2] = 0
    # apply_profile_boundaries
    mapped = terrain_solid = 0
    fl3[0, :, :] = TYPE_S; u3[:, 0, :, :] = 0
    bmask = np.zeros((Nz, Ny, Nx), bool)
    bmask[:, :, 0] = bmask[:, :, Nx - 1] = bmask[:, 0, :] = bmask[:, Ny - 1, :] = True
    bmask[Nz - 1, :, :] = True
    bmask[0, :, :] = False
    for z in range(1, Nz):
        m = bmask[z] & ((fl3[z] & TYPE_S) == 0)
        if not m.any(): continue
        if posz[z] <= flat_ground:
            fl3[z][m] = TYPE_S; u3[:, z][:, m] = 0; terrain_solid += int(m.sum()); continue
        fl3[z][m] |= TYPE_E
        side = np.zeros((Ny, Nx), bool); side[:, 0] = side[:, Nx - 1] = side[0, :] = side[Ny - 1, :] = True
        s_here = spd[z]
        s_cap = speed(posz[side_ref_z_cap], flat_ground) if (side_ref_z_cap >= 0 and z > side_ref_z_cap) else s_here
        val = np.where(side, s_cap, s_here).astype(f32)
        u3[0, z][m] = (dir_x * val)[m]; u3[1, z][m] = (dir_y * val)[m]; u3[2, z][m] = 0
        mapped += int(m.sum())
    return dict(case=case, tri_lattice=tri, Nx=Nx, Ny=Ny, Nz=Nz, Nz_core=Nz_core, D=D, nu=lbm_nu, units=units, cell_m=cell_m,
                si_ref_u=si_ref_u, u_scale=u_scale, flags=flags, u=u, rho=rho,
                buffer_active=buffer_active, buffer_N=nbuf, buffer_inv_tau=buffer_inv_tau,
                buffer_nudge_vertical=buffer_nudge_vertical, buffer_face=buffer_face,
                sponge_active=sponge_active, sponge_N=ns, sponge_inv_tau=sponge_inv_tau,
                side_ref_z_cap=side_ref_z_cap, n_solid_vox=n_solid_vox, mapped_bc=mapped,
                terrain_solid_bc=terrain_solid, run_nstep=run_nstep, dir=(dir_x, dir_y), downstream_bc=dbc,
                scale_geom=scale_geom, vtk_origin_shift=vtk_origin_shift, table_top=table_top,
                core_top_si=core_top_si, solver_top_si=solver_top_si, prof_si=np.array(prof_si, f32),
                si_u_factor=units.si_u(f32(1.0)), si_rho_factor=units.si_rho(f32(1.0)))
