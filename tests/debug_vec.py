import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import latticeurbanwind_amd as luw
from latticeurbanwind_amd import capi
from oracle import oracle
from helpers import synthetic_state
Nx, Ny, Nz = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 32, 32))]
shell = sys.argv[4] if len(sys.argv) > 4 else "luw"
st = synthetic_state(Nx, Ny, Nz, seed=3, shell=None if shell == "none" else shell)
for kern in (capi.KERNEL_SCALAR, capi.KERNEL_VEC4):
    g = luw.LBM(Nx, Ny, Nz, 1e-4, kernel=kern, update_fields_every_step=True)
    o = oracle.OracleLBM(Nx, Ny, Nz, 1e-4)
    g.flags.data[:] = st[0]; g.u.data[:] = st[1]; g.rho.data[:] = st[2]
    o.flags[:] = st[0]; o.u[:] = st[1]; o.rho[:] = st[2]
    g.run(0); o.initialize()
    for step in range(3):
        g.run(1); o.run(1)
        fi = g.download_fi().reshape(19, Nz, Ny, Nx); ref = o.fi.reshape(19, Nz, Ny, Nx)
        bad = fi != ref
        print("kernel", kern, "step", step + 1, "bad", int(bad.sum()))
        if bad.any():
            print(" per plane:", [int(bad[i].sum()) for i in range(19)])
            idx = np.argwhere(bad)
            print(" x hist:", np.bincount(idx[:, 3], minlength=Nx).tolist())
            print(" first:", idx[:8].tolist())
            i, z, y, x = idx[0]
            print(" got", fi[i, z, y, x], "ref", ref[i, z, y, x], "flags", st[0].reshape(Nz, Ny, Nx)[z, y, max(0, x - 2):x + 3].tolist())
            d = np.abs(fi - ref)[bad]
            print(" max abs diff", d.max(), "median", np.median(d))
            break
