"""Shared test helpers: synthetic states applied identically to the oracle (oracle.OracleLBM) and to the product
(latticeurbanwind_amd.LBM), which expose the same host arrays (rho, u, flags, F) in the reference's layout."""
import numpy as np

TYPE_S, TYPE_E, TYPE_T = 0x01, 0x02, 0x04


def synthetic_state(Nx, Ny, Nz, seed=1, solids=True, shell="E", u0=0.05):
    """Deterministic 'urban' state: smooth shear flow + seeded perturbation, optional box solids, outer shell of
    TYPE_E ("E"), solid ground + TYPE_E elsewhere ("luw", like FX/setup.cpp:5945-5985) or fully periodic (None)."""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.arange(Nz), np.arange(Ny), np.arange(Nx), indexing="ij")
    flags = np.zeros((Nz, Ny, Nx), np.uint8)
    if solids:
        bx0, bx1 = Nx // 3, max(Nx // 3 + 1, Nx // 2)
        by0, by1 = Ny // 4, max(Ny // 4 + 1, Ny // 2)
        flags[: max(1, Nz // 2), by0:by1, bx0:bx1] = TYPE_S
        if Nx > 8 and Ny > 8:
            flags[: max(1, Nz // 3), Ny - 4:Ny - 2, 2:5] = TYPE_S
    ux = (u0 * (0.3 + 0.7 * (z + 0.5) / Nz) + 0.01 * rng.standard_normal((Nz, Ny, Nx))).astype(np.float32)
    uy = (0.02 * np.sin(2 * np.pi * x / max(Nx, 2)) + 0.01 * rng.standard_normal((Nz, Ny, Nx))).astype(np.float32)
    uz = (0.01 * rng.standard_normal((Nz, Ny, Nx))).astype(np.float32)
    rho = (1.0 + 0.01 * rng.standard_normal((Nz, Ny, Nx))).astype(np.float32)
    if shell in ("E", "luw"):
        b = np.zeros((Nz, Ny, Nx), bool)
        b[0], b[-1], b[:, 0], b[:, -1], b[:, :, 0], b[:, :, -1] = True, True, True, True, True, True
        if shell == "luw":
            flags[0] = TYPE_S
            b[0] = False
        m = b & (flags != TYPE_S)
        flags[m] |= TYPE_E
        rho[m] = 1.0
    s = (flags & TYPE_S) != 0
    ux[s] = 0; uy[s] = 0; uz[s] = 0
    return flags.ravel(), np.concatenate([ux.ravel(), uy.ravel(), uz.ravel()]), rho.ravel()


def apply_state(lbm, flags, u, rho, F=None):
    lbm.flags.data[:] = flags if hasattr(lbm.flags, "data") else 0
    lbm.u.data[:] = u
    lbm.rho.data[:] = rho
    if F is not None:
        lbm.F.data[:] = F


def apply_state_oracle(o, flags, u, rho, F=None):
    o.flags[:] = flags
    o.u[:] = u
    o.rho[:] = rho
    if F is not None:
        o.F[:] = F


def rmse_u(ua, ub, fluid_mask):
    d = (ua.reshape(3, -1) - ub.reshape(3, -1))[:, fluid_mask]
    return float(np.sqrt((d.astype(np.float64) ** 2).sum(0).mean()))


def thermal_state(flags, gN, seed=5):
    """TYPE_T presets for thermal tests: 3 % of the non-solid cells become heat sources with T in 1 +- 0.05; T = 1 elsewhere"""
    rng = np.random.default_rng(seed)
    f = flags.copy()
    pick = (rng.random(f.size) < 0.03) & ((f & TYPE_S) == 0)
    f[pick] |= TYPE_T
    T = np.ones(f.size, np.float32)
    T[pick] = (1.0 + 0.05 * rng.standard_normal(int(pick.sum()))).astype(np.float32)
    return f, T


# ---- tolerance gates tied to what was observed
# The distance of this path from the REAL reference's fields is whatever the reference's own (not bit-defined) arithmetic leaves: 0.5-1.3e-7 with FP32 DDFs,
# 0.3-3e-5 after 64 LES steps with FP16C storage.  Class-level ceilings alone (1e-6 / 1e-4) would let a regression of an order of
# magnitude pass, so every comparison against a reference fixture is ALSO held to twice the value recorded in tests/golden/observed_rmse.json (the product
# equals the CPU oracle bit for bit, so the recorded values reproduce exactly).  LUW_RECORD_RMSE=<file>: append what is observed instead (new fixtures).
import json as _json
import os as _os

_OBSERVED = None


def observed_table():
    global _OBSERVED
    if _OBSERVED is None:
        path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "observed_rmse.json")
        _OBSERVED = _json.load(open(path)) if _os.path.exists(path) else {}
    return _OBSERVED


def check_gate(key, value, ceiling, what="u RMSE"):
    """value < ceiling (the class-level gate) and value <= 2 x the recorded observation of `key`"""
    rec = _os.environ.get("LUW_RECORD_RMSE")
    if rec:
        with open(rec, "a") as f:
            f.write(_json.dumps({key: value}) + "\n")
    assert value < ceiling, "%s %.3e (%s) above the ceiling %.1e" % (what, value, key, ceiling)
    seen = observed_table().get(key)
    if seen is not None and not rec:
        assert value <= 2.0 * seen + 1e-12, "%s %.3e (%s) is more than twice the recorded %.3e" % (what, value, key, seen)
    return value
