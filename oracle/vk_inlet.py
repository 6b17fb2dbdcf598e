"""CPU restatement of the reference's von-Karman inlet HOST logic (VonKarmanInletUpdater, FX/setup.cpp:413-1149) for one
domain: inlet-cell selection (:667-765), random Fourier modes of the von-Karman spectrum (:777-850, std::mt19937_64 +
std::uniform_real_distribution<float> restated from the C++ standard / libstdc++ generate_canonical), table packing
(:886-1057) and the per-step time parameters (:1118-1140).  The device kernel's restatement is luwo_vk_inlet_apply in
luw_oracle.c.  TEST INFRASTRUCTURE ONLY."""
import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
_m = ctypes.CDLL(ctypes.util.find_library("m"))
for _n in ("logf", "expf", "cosf", "sinf", "sqrtf"):
    getattr(_m, _n).argtypes = [ctypes.c_float]; getattr(_m, _n).restype = ctypes.c_float
_m.powf.argtypes = [ctypes.c_float, ctypes.c_float]; _m.powf.restype = ctypes.c_float
logf = lambda x: f32(_m.logf(float(x))); expf = lambda x: f32(_m.expf(float(x))); cosf = lambda x: f32(_m.cosf(float(x)))
sinf = lambda x: f32(_m.sinf(float(x))); sqrtf = lambda x: f32(_m.sqrtf(float(x))); powf = lambda x, y: f32(_m.powf(float(x), float(y)))
PIF = f32(3.1415927)


class MT19937_64:
    """std::mt19937_64 (ISO C++ [rand.predef]: w=64 n=312 m=156 r=31 a=0xb5026f5aa96619e9 u=29 d=0x5555555555555555
    s=17 b=0x71d67fffeda60000 t=37 c=0xfff7eee000000000 l=43 f=6364136223846793005)"""
    M64 = (1 << 64) - 1

    def __init__(self, seed):
        mt = [0] * 312
        mt[0] = seed & self.M64
        for i in range(1, 312):
            mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & self.M64
        self.mt, self.i = mt, 312

    def __call__(self):
        if self.i >= 312:
            mt = self.mt
            for k in range(312):
                x = (mt[k] & 0xFFFFFFFF80000000) | (mt[(k + 1) % 312] & 0x7FFFFFFF)
                xa = x >> 1
                if x & 1:
                    xa ^= 0xB5026F5AA96619E9
                mt[k] = mt[(k + 156) % 312] ^ xa
            self.i = 0
        y = self.mt[self.i]; self.i += 1
        y ^= (y >> 29) & 0x5555555555555555
        y ^= (y << 17) & 0x71D67FFFEDA60000
        y ^= (y << 37) & 0xFFF7EEE000000000
        y ^= y >> 43
        return y & self.M64


def uni01(rng):
    """std::uniform_real_distribution<float>(0,1)(rng) with libstdc++: generate_canonical<float,24>: one 64-bit draw,
    float(draw)/float(2^64), clamped below 1"""
    r = f32(float(rng())) / f32(18446744073709551616.0)
    if r >= f32(1.0):
        r = np.nextafter(f32(1.0), f32(0.0))
    return f32(r)


def build_modes_for_seed(L, nmodes, u_ref, conv_dir, seed, aniso=(1.0, 1.0, 1.0)):
    L = f32(L)
    k_max = PIF / f32(1.0)
    k_min = f32(2.0) * PIF / (f32(10.0) * L)
    if not (k_min > 0) or not np.isfinite(k_min): k_min = f32(1.0e-4)
    if k_min >= f32(0.99) * k_max: k_min = f32(0.1) * k_max
    log_k_min, log_k_max = logf(k_min), logf(k_max)
    span = max(log_k_max - log_k_min, f32(1.0e-6))
    rng = MT19937_64(seed)
    a_raw, modes, sum_a2 = [], [], 0.0
    for m in range(nmodes):
        xi = (f32(m) + uni01(rng)) / f32(nmodes)
        k = expf(log_k_min + xi * span)
        zeta = f32(2.0) * uni01(rng) - f32(1.0)
        az = f32(2.0) * PIF * uni01(rng)
        r = sqrtf(max(f32(0.0), f32(1.0) - zeta * zeta))
        dx, dy, dz = r * cosf(az), r * sinf(az), zeta
        kx, ky, kz = k * dx, k * dy, k * dz
        kL = k * L
        denom = powf(f32(1.0) + kL * kL, f32(17.0) / f32(6.0))
        W = powf(k, f32(4.0)) / denom if denom > 0 else f32(0.0)
        a = sqrtf(max(W, f32(0.0)))
        a_raw.append(a); sum_a2 += float(a) * float(a)
        omega = f32(u_ref) * (kx * f32(conv_dir[0]) + ky * f32(conv_dir[1]) + kz * f32(conv_dir[2]))
        phix = f32(2.0) * PIF * uni01(rng); phiy = f32(2.0) * PIF * uni01(rng); phiz = f32(2.0) * PIF * uni01(rng)
        modes.append([kx, ky, kz, omega, f32(0), f32(0), f32(0), phix, phiy, phiz])
    var = 0.5 * sum_a2
    scale = f32(1.0) / f32(np.sqrt(var))
    for m in range(nmodes):
        A = a_raw[m] * scale
        modes[m][4], modes[m][5], modes[m][6] = A * f32(aniso[0]), A * f32(aniso[1]), A * f32(aniso[2])
    return np.array(modes, f32)          # (M, 10)


FACE_N = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, -1]], f32)


def build_tables(Nx, Ny, Nz, flags, u, ti=0.05, sigma_lbm=0.0, L_lbm=100.0, nmodes=256, seed=100, face_mode="ALL_SIDES", downstream_face_id=-1,
                 same_realization=True):
    """default deck behaviour: face_mode AUTO_SIDES with inflow_only=false -> ALL_SIDES, uc_mode NORM_MEAN"""
    N = Nx * Ny * Nz
    pts = [[] for _ in range(5)]

    def valid(n, fid, z):
        if z == 0 or (flags[n] & 1) or not (flags[n] & 2): return False
        if face_mode == "ALL_SIDES": return fid != 4
        if face_mode == "EXCLUDE_DOWNSTREAM_SIDES": return fid != 4 and not (downstream_face_id >= 0 and fid == downstream_face_id)
        if face_mode == "EXCLUDE_DOWNSTREAM": return not (downstream_face_id >= 0 and fid == downstream_face_id)
        return True

    def add(fid, x, y, z):
        n = x + (y + z * Ny) * Nx
        if valid(n, fid, z): pts[fid].append((n, x, y, z, f32(u[n]), f32(u[N + n]), f32(u[2 * N + n])))
    for z in range(1, Nz - 1):
        for y in range(Ny): add(0, 0, y, z); add(1, Nx - 1, y, z)
        for x in range(1, Nx - 1): add(2, x, 0, z); add(3, x, Ny - 1, z)
    for y in range(Ny):
        for x in range(Nx): add(4, x, y, Nz - 1)
    enabled, mean_all, sum_mag, cnt = [False] * 5, np.zeros(3, f32), 0.0, 0
    for f in range(5):
        if not pts[f]: continue
        mu = np.zeros(3, f32)
        for p in pts[f]: mu = (mu + np.array(p[4:7], f32)).astype(f32)
        mu = (mu / f32(len(pts[f]))).astype(f32)
        uc = sqrtf(mu[0] * mu[0] + mu[1] * mu[1] + mu[2] * mu[2])
        enabled[f] = bool(uc > f32(1.0e-7))
    for f in range(5):
        if not enabled[f]: continue
        for p in pts[f]:
            b = np.array(p[4:7], f32)
            mean_all = (mean_all + b).astype(f32)
            sum_mag += float(sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2])); cnt += 1
    if cnt == 0: return None
    u_ref = f32(sum_mag / cnt)
    conv = (mean_all / f32(cnt)).astype(f32)
    cl = sqrtf(conv[0] * conv[0] + conv[1] * conv[1] + conv[2] * conv[2])
    conv = (conv / cl).astype(f32) if cl > f32(1.0e-7) else np.array([1, 0, 0], f32)
    M = nmodes
    mode_data = np.zeros((10, 5 * M), f32)
    shared = build_modes_for_seed(L_lbm, M, u_ref, conv, seed) if same_realization else None
    for f in range(5):
        if enabled[f]:
            mode_data[:, f * M:(f + 1) * M] = shared.T
    cell, face, pdata = [], [], []
    for f in range(5):
        if not enabled[f]: continue
        for p in pts[f]:
            b = np.array(p[4:7], f32)
            uch = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2])
            sig = f32(ti) * uch if ti > 0 else f32(sigma_lbm)
            if not (sig > 0): continue
            cell.append(p[0]); face.append(f); pdata.append([f32(p[1]), f32(p[2]), f32(p[3]), b[0], b[1], b[2], sig])
    pd = np.array(pdata, f32).T.copy()
    return dict(point_cell=np.array(cell, np.uint64), point_face=np.array(face, np.uint8), point_data=pd.ravel(), mode_data=mode_data.ravel(),
                point_count=len(cell), mode_count=M, u_ref=u_ref, conv=conv)


def time_params(t, update_stride=1, stride_interpolation=False):
    stride = update_stride if update_stride > 1 else 1
    if stride <= 1: return 0, f32(t), f32(t), f32(0)
    anchor = (t // stride) * stride
    if stride_interpolation: return 1, f32(anchor), f32(anchor + stride), f32(t - anchor) / f32(stride)
    return 0, f32(anchor), f32(anchor), f32(0)
